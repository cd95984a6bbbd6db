// placeholder until the persistent solver lands
#include "pdp_device.hpp"
extern "C" int pdp_sp_solve(pdp_problem *p, pdp_solve_args *args, void *stream)
{
    (void)p; (void)args; (void)stream;
    pdp_set_error("pdp_sp_solve: not built yet");
    return PDP_ERR_UNSUPPORTED;
}
