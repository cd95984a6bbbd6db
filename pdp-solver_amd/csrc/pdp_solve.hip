// pdp_solve.hip -- persistent solver: the PDP iteration loop without a host round trip per iteration.
// replaces: PropagatorDecimatorSolverBase._forward_core (reference: src/pdp/nn/solver.py:355-386) for the
// classical triple SurveyPropagator + SequentialDecimator(SurveyScorer) + IdentityPredictor together with the
// per-iteration termination check (src/pdp/trainer.py:150-162).
//
// One workgroup owns one CNF instance.  When the instance fits (the common case: the BASELINE configs have <= ~5k
// edges per instance) its topology and its whole message state live in LDS for a chunk of iterations per launch
// (k_sp_solve_lds); between launches it lives in a private slot-major record in HBM, and all launches of a call plus the
// device-side decisions between them are enqueued up front (sp_solve_resident).  Larger instances run the same algorithm
// on HBM-resident arrays (k_sp_solve, host-driven chunk loop).  DESIGN.md section 4.2 is the narrative.
//
// Cross-instance couplings of the reference cannot be honoured inside independent
// workgroups, so the kernel SPECULATES that they are inert -- batch-global min of each arg-max/max operand is 0
// and no NaN appears -- and records per (iteration, call site) whether some instance really had an exact zero.
// The host verifies the record; on a miss the call returns PDP_ERR_SPECULATION and the caller reruns the batch
// through the strict step-wise entry points (pdp_ops.hip), so results always equal the reference semantics.
#include "pdp_device.hpp"
#include <stdlib.h>
#include <algorithm>
#include <vector>
#include <type_traits>

#define ST(s) ((hipStream_t)(s))
#define PDP_RETRY_WITH_FORCE (-77)      // internal: sp_solve_resident -> sp_solve_speculative (never leaves the library)

// instance view used by the solver: same field names as Inst, index type templated (u16 in LDS, i32 in HBM)
template <class IT>
struct SView {
    int b, n, m, e;
    const IT *e_var, *e_fn, *v_ptr, *v_edges, *f_ptr, *f_edges;
    const int8_t *sgn;
    float *av, *af, *sol;
    float *emask;           // [e]
    float *qu; int qstride; // q[:,0]
    float *eta; int estride;// fs[:,0]
    const float *force; int fstride;   // fs[:,1]
    float *s0, *s1, *s2, *s3;           // [e] scratch
    float *S;               // [m]
    float *P, *N, *xv1, *xv2, *score, *coeff, *assign;   // [n]
    int32_t *deg, *sdeg;    // [n]
    uint8_t *flag_v, *flag_f, *flag_f2;
};

struct SolveParams {
    int T;
    float tol, t_max, pi;
    float *q, *fs;              // [E,3], [E,2]
    uint8_t *amask;             // [B]
    float *prev, *counters;     // decimator state
    int has_prev;
    int check_termination;
    int has_edge_mask;          // problem->has_edge_mask at entry
    int final_chunk;            // this launch ends the loop: rebuild q_s / q_dc for instances that are still active
    int poison_from;            // first iteration at which the batch is NaN-poisoned (INT_MAX: never); pass 2 only
    const int32_t *inst_list;   // replay pass: instances to run (grid = list length); NULL: instance = blockIdx.x
    int32_t *last_event;        // [B] pass 1: last iteration with a gate / convergence event (-1: none)
    // state the instance is LOADED from (pass 1: the live arrays, replay: the snapshot taken before pass 1)
    const float *src_q, *src_fs, *src_av, *src_af, *src_sol, *src_sat, *src_emask, *src_prev, *src_cnt;
    const uint8_t *src_amask;
    uint32_t *nan_iter;         // device word: min over instances of the first iteration whose surveys contain a NaN
    uint32_t *spec_zero;        // [T] bits: site0 (survey max), site1 (diff max), site2 (coeff argmax): some instance had an exact 0
    uint32_t *spec_used;        // [T] bits: some instance evaluated the site
    // HBM-mode scratch
    float *ws_e[4]; float *ws_f; float *ws_v[7]; int32_t *ws_vi[3]; uint8_t *ws_fu[2];
    float *ws_r;                // [E][4] per-edge record of the sweep: survey | exp(30 d) | d | sign * (1: edge mask on, 2: off)
    // LDS-resident kernel: private instance records (see BlobLayout) and device-side control
    int chunk_start;            // iterations completed before this launch
    struct SolveCtl *ctl;       // this chunk's control block
    struct SolveCall *call;     // the call's control block
    const char *stat;           // static (topology) records
    const char *dyn_in;         // dynamic records this launch resumes from
    char *dyn_out;              // dynamic records this launch leaves behind
    const int64_t *stat_off, *dyn_off;   // [B] byte offsets of the instance records
    const float *prev_slots;    // [E] slot-major copy of the decimator's previous surveys (first launch of a call only)
    int debug_skip;             // -DPDP_PHASE_PROF builds: bit mask of phases to skip (timing experiments, results are wrong)
    const float *frc_in;        // Reinforce: [E] slot-major force column this launch resumes from (double-buffered like the dynamic records:
    float *frc_out;             //            a poison replay must not see what pass 1 left behind)
    const float *coins;         // Reinforce: [T of the call] the shared coin of every iteration (pdp_decimate.py:218), drawn by the caller
    float dprob;                // Reinforce: decimation probability
    // per-instance routing (mixed batches): the LDS-resident kernel runs the instances of fit_list, the HBM-resident kernel those of
    // big_list, both under the device-side control blocks of the chunk
    const int32_t *fit_list;    // NULL: instance = blockIdx.x
    const int32_t *big_list;    // HBM-resident kernel: NULL: instance = blockIdx.x
    int hbm_device_ctl;         // HBM-resident kernel: 1 = stop / poison / replay decisions come from ctl / call (else from the host: poison_from)
    uint32_t spin_limit;        // team / lock-step barriers: polls before a wait gives up (0: the default; pdp_spin_limit())
    int lock_size;              // lock-step launch: workgroups the batch barrier waits for (0: the grid; a test makes it one too many)
    int hbm_replay;             // HBM-resident kernel under device control: this launch is the replay pass
    uint32_t *w_perm_zero, *w_iters_run, *w_violation;   // HBM-resident kernel: where its control words go (pv.flags slots, or the chunk's SolveCtl)
    // teams (k_sp_solve<NT, true>): team_size workgroups per instance, team_count instances, numbered slot-minor over team_slots
    int team_size, team_count, team_slots;
    int team_no_xcd;            // debugging: always take the agent-scope barrier
    int exact;                  // HBM-resident kernel, single-instance batch: the batch-global minima ARE the instance's own, nothing is speculated
    int rf;                     // HBM-resident kernel: the Reinforce triple (coins, dprob as for the LDS-resident kernel; tol = the gate's 0.01)
    int isolate;                // isolated instances (a NaN stays inside its instance, pass 1 is final)
    uint32_t *risk;             // [B] LDS-resident pass 1: bits of the smallest q normalisation of the launch's last sweep (see k_solve_finish); NULL: off
    int no_scorer_reuse;        // PDP_SOLVE_NO_SCORER_REUSE=1: the decimation's scorer always takes its own logs (A/B switch)
    int no_event_look;          // PDP_SOLVE_NO_EVENT_LOOK=1: pass-1 workgroups look for a recorded NaN sweep only when they start
    int rf_no_fused_step;       // PDP_SOLVE_RF_NO_FUSED_STEP=1: a Reinforce coin sweep runs E2's plain form and its step uses X / Y as scratch (round 5's form)
    int debug_ghost_inject;     // tests: see the exit path of k_sp_solve_lds
    int adopt_poison;           // LDS-resident pass 1: take a first-NaN sweep other workgroups of the launch already recorded (PDP_SOLVE_NO_ADOPT=1: off)
    int lds_tickets;            // LDS-resident kernel, pass 1: 0 = instance blockIdx.x, else the number of instances the workgroups draw tickets for
    uint8_t *ghost_flag;        // LDS-resident kernel: [B] 2 = an instance that left inactive with iterations to come failed its ghost sweep (lds_ghost_bad)
    uint32_t *team_ws;          // [team_count][PDP_TEAM_WORDS], zeroed before every launch
};

// Device-side control of the chunked persistent solve: the host enqueues every launch of a call up front and reads
// one SolveCall back at the end.
struct SolveCtl {               // one per chunk
    uint32_t nan_iter;          // min over instances of the first chunk-relative iteration with a NaN survey (0xffffffff: none)
    uint32_t perm_zero;         // first iteration from which an exited instance guarantees exact zeros (0xffffffff: none)
    uint32_t violation;         // a speculation the kernel itself can see failed
    uint32_t iters_run;         // max over instances of the iterations run in this chunk
    uint32_t replay_count;      // instances listed for the poison replay
    uint32_t do_replay;
    int32_t poison_from;        // replay pass: first poisoned (chunk-relative) iteration
    uint32_t ticket;            // next instance of the LDS-resident pass (launches with sp.lds_tickets)
};
struct SolveCall {
    uint32_t poisoned_all;      // a NaN poisoned the batch in an earlier chunk: later chunks run poisoned from their first iteration
    uint32_t stop;              // every instance went inactive (solver.py:383): the remaining launches return immediately
    uint32_t fail;              // a cross-instance coupling became active: the caller must rerun step-wise
    uint32_t total_iters;
    uint32_t force_seen;        // k_solve_import found a non-zero external force while the call runs the force-free instantiation: every
                                // solver launch returns at once, nothing is modified, the host reruns the call with the force column
    uint32_t pad[3];
};

// Private instance records of the LDS-resident solver.  Between two launches of a call an instance lives in HBM as the
// verbatim image of its LDS arrays (slot-major, 16-byte aligned pieces): resuming is a coalesced copy instead of a
// gather through the CSR, and the record a launch resumed from stays intact, which is the snapshot the poison replay needs.
//   static  (per problem): pvv | e2p | v_ptr | f_ptr | vord (variables by descending degree)
//   dynamic (two copies, ping-pong): header | QU | E | pcc | af | av | sol
struct DynHeader { uint32_t active, done, perm_zero, simplified; float cnt, is_sat; int32_t nsat_p1; float pad2; };   // nsat_p1: satisfied clauses under `sol` + 1 (0: not counted yet)
//   // simplified: 0 unknown, 1 the state is a simplify() fix-point, 2 it is not
struct BlobLayout { size_t pvv, e2p, vptr, fptr, vord, stat_bytes, hdr, QU, E, pcc, af, av, sol, dyn_bytes; };
__host__ __device__ inline BlobLayout blob_layout(int n, int m, int ne)
{
    auto a16 = [](size_t x) { return (x + 15) & ~(size_t)15; };
    BlobLayout b; size_t o = 0;
    b.pvv = o; o += a16((size_t)ne * 2); b.e2p = o; o += a16((size_t)ne * 2);
    b.vptr = o; o += a16((size_t)(n + 1) * 2); b.fptr = o; o += a16((size_t)(m + 1) * 2);
    b.vord = o; o += a16((size_t)n * 2); b.stat_bytes = o;
    o = 0;
    b.hdr = o; o += a16(sizeof(DynHeader));
    b.QU = o; o += a16((size_t)ne * 4); b.E = o; o += a16((size_t)ne * 4); b.pcc = o; o += a16((size_t)ne * 2);
    b.af = o; o += a16((size_t)m * 4); b.av = o; o += a16((size_t)n * 4); b.sol = o; o += a16((size_t)n * 4); b.dyn_bytes = o;
    return b;
}

__device__ __forceinline__ size_t align16(size_t x) { return (x + 15) & ~(size_t)15; }

template <class T>
__device__ __forceinline__ T *carve(unsigned char *&p, size_t count)
{
    T *r = reinterpret_cast<T *>(p);
    p += align16(count * sizeof(T));
    return r;
}

// (TeamView = Teamed<SView<int32_t>>: pdp_common.hpp holds the team machinery -- barrier, mailboxes, reductions)
typedef Teamed<SView<int32_t>> TeamView;

// the reductions that close an iteration of the sweep -- two maxima (NaN is maximal), a bit mask and, in exact mode, two minima --
// on one pair of workgroup barriers
struct IterRed { float a, b, mna, mnb; int bits; };
struct OpMinLess { __device__ float operator()(float a, float b) const { return b < a ? b : a; } };
__device__ __forceinline__ IterRed block_iter_reduce(IterRed x, bool with_min, float *s /*[5 * waves]*/)
{
    const int lane = threadIdx.x & (PDP_WAVE - 1), wid = threadIdx.x / PDP_WAVE;
    const int nw = (blockDim.x + PDP_WAVE - 1) / PDP_WAVE;
    x.a = wave_reduce(x.a, OpMaxNan(), -PDP_INF);
    x.b = wave_reduce(x.b, OpMaxNan(), -PDP_INF);
    x.bits = wave_reduce(x.bits, OpOrI(), 0);
    if (with_min) { x.mna = wave_reduce(x.mna, OpMinLess(), PDP_INF); x.mnb = wave_reduce(x.mnb, OpMinLess(), PDP_INF); }
    if (lane == PDP_WAVE - 1) { float *w = s + 5 * wid; w[0] = x.a; w[1] = x.b; w[2] = __int_as_float(x.bits); w[3] = x.mna; w[4] = x.mnb; }
    __syncthreads();
    IterRed r; r.a = -PDP_INF; r.b = -PDP_INF; r.mna = PDP_INF; r.mnb = PDP_INF; r.bits = 0;
    for (int i = 0; i < nw; ++i) {
        const float *w = s + 5 * i;
        r.a = pdp_max(r.a, w[0]); r.b = pdp_max(r.b, w[1]); r.bits |= __float_as_int(w[2]);
        if (with_min) { r.mna = OpMinLess()(r.mna, w[3]); r.mnb = OpMinLess()(r.mnb, w[4]); }
    }
    __syncthreads();
    return r;
}
template <class IT>
__device__ __forceinline__ IterRed team_iter_reduce(const SView<IT> &, IterRed x, bool with_min, float *red5) { return block_iter_reduce(x, with_min, red5); }
template <class B>
__device__ __forceinline__ IterRed team_iter_reduce(const Teamed<B> &t, IterRed x, bool with_min, float *red5)
{
    x = block_iter_reduce(x, with_min, red5);
    if (t.size == 1) return x;
    uint32_t *box = team_box(t);
    if (threadIdx.x == 0) {
        uint32_t *mine = &box[PDP_BOX_WORDS * t.rank];
        box_put(&mine[0], __float_as_uint(x.a)); box_put(&mine[1], __float_as_uint(x.b)); box_put(&mine[2], (uint32_t)x.bits);
        if (with_min) { box_put(&mine[3], __float_as_uint(x.mna)); box_put(&mine[4], __float_as_uint(x.mnb)); }
    }
    team_sync(t);
    IterRed y; y.a = -PDP_INF; y.b = -PDP_INF; y.mna = PDP_INF; y.mnb = PDP_INF; y.bits = 0;
    for (int r = (int)threadIdx.x; r < t.size; r += (int)blockDim.x) {
        const uint32_t *w = &box[PDP_BOX_WORDS * r];
        y.a = pdp_max(y.a, __uint_as_float(box_get(&w[0]))); y.b = pdp_max(y.b, __uint_as_float(box_get(&w[1]))); y.bits |= (int)box_get(&w[2]);
        if (with_min) { y.mna = OpMinLess()(y.mna, __uint_as_float(box_get(&w[3]))); y.mnb = OpMinLess()(y.mnb, __uint_as_float(box_get(&w[4]))); }
    }
    return block_iter_reduce(y, with_min, red5);
}

// Reinforce triple on the HBM-resident view (the cold part of an iteration; lds_reinforce_step is the LDS-resident twin): with `do_force`
// the force update of ReinforceDecimator.forward (pdp_decimate.py:218-232: SurveyScorer on the new surveys and the OLD force, then
// force <- sign(score) on every edge of the instance), always ReinforcePredictor (pdp_predict.py:221-226) + _update_solution
// (solver.py:388-399).  The force a sweep read is kept in s2 (free after P4): the write-back rebuilds q_s / q_dc of the last sweep from it.
// Returns 1 if a score was NaN.
template <class V>
__device__ int hbm_reinforce_step(const V &I, float *fs /*[e][2]*/, float pi, int do_force)
{
    const int tid = team_tid(I), nt = team_nt(I);
    if (do_force) {
        for (int e = tid; e < I.e; e += nt) I.s3[e] = pdp_safe_log(1.0f - I.eta[e * I.estride], PDP_SCORER_EPS) * (0.0f + I.af[I.e_fn[e]]);
        team_sync(I);
    }
    int bad = 0;
    for (int v = tid; v < I.n; v += nt) {
        const int a = I.v_ptr[v], bnd = I.v_ptr[v + 1];
        float ext = 0.0f;
        if (do_force) {
            float pos = 0.0f, neg = 0.0f, all = 0.0f;
            for (int k = a; k < bnd; ++k) {
                const int e = I.v_edges[k];
                const float f = I.s3[e];
                const int sg = I.sgn[e];
                ext = ext + fs[2 * e + 1];
                pos = pos + ((sg == 1) ? 1.0f : 0.0f) * f;
                neg = neg + ((sg == -1) ? 1.0f : 0.0f) * f;
                all = all + f;
            }
            const float sc = d_score_from_sums(pos, neg, all, ext, pi);
            if (sc != sc) bad = 1;
            const float sg = 0.0f + pdp_sign(sc);                   // torch.sign(NaN) is 0
            // mask * sign + (1 - mask) * old with mask == 1 (the instance is active, old is finite)
            for (int k = a; k < bnd; ++k) { const int e = I.v_edges[k]; I.s2[e] = fs[2 * e + 1]; fs[2 * e + 1] = sg; }
        }
        ext = 0.0f;
        for (int k = a; k < bnd; ++k) ext = ext + fs[2 * I.v_edges[k] + 1];
        const float pred = (ext > 0.0f) ? 1.0f : 0.0f;
        const float av = I.av[v];
        if (av == 1.0f) I.sol[v] = av * pred + (1.0f - av) * I.sol[v];      // only active variables take the prediction (solver.py:395-397)
    }
    return team_any(I, bad);
}

// Ghost sweep of an inactive instance on an HBM-resident view: would its next sweep, or the scorer on its frozen surveys, produce a
// non-finite value?  (k_sp_solve explains why that matters.)  Clobbers s0, s3, S, score, coeff.  Team-aware.
template <class V>
__device__ int d_ghost_bad(const V &I, float pi, float L0h, float L1h, bool use_em, bool check_score)
{
    const int tid = team_tid(I), nt = team_nt(I), n = I.n, m = I.m, ne = I.e;
        int bad = 0;
        for (int c = tid; c < m; c += nt) {
            float acc = 0.0f;
            for (int k = I.f_ptr[c]; k < I.f_ptr[c + 1]; ++k) {
                const int e = I.f_edges[k];
                float x = pdp_safe_log(I.qu[e * I.qstride], PDP_SP_EPS);
                if (use_em) x = x * I.emask[e];
                I.s0[e] = x; acc = acc + x;
            }
            I.S[c] = acc;
        }
        for (int v = tid; v < n; v += nt) {
            float P = 0.0f, N = 0.0f;
            for (int k = I.v_ptr[v]; k < I.v_ptr[v + 1]; ++k) {
                const int e = I.v_edges[k];
                float y = pdp_safe_log(1.0f - I.eta[e * I.estride], PDP_SP_EPS);
                if (use_em) y = y * I.emask[e];
                I.s3[e] = y;
                const int sg = I.sgn[e];
                P = P + ((sg == 1) ? 1.0f : 0.0f) * y;
                N = N + ((sg == -1) ? 1.0f : 0.0f) * y;
            }
            I.score[v] = P; I.coeff[v] = N;
        }
        team_sync(I);
        for (int e = tid; e < ne; e += nt) {
            const float agg = (0.0f + I.S[I.e_fn[e]]) - I.s0[e];
            const float eta_new = pdp_safe_exp(agg);
            const int v = I.e_var[e];
            const SpOut o = d_sp_edge((float)I.sgn[e], I.score[v], I.coeff[v], I.s3[e], I.force[e * I.fstride], L0h, L1h);
            if (!pdp_finite(eta_new) || !pdp_finite(o.qu) || !pdp_finite(o.qs) || !pdp_finite(o.dc)) {
                bad = 1;
#ifdef PDP_PHASE_PROF
                printf("[ghost] inst %d edge %d: eta_new %g qu %g qs %g dc %g | P %g N %g y %g S %g x %g eta %g q_u %g\n", I.b, e, eta_new, o.qu, o.qs, o.dc, I.score[v], I.coeff[v], I.s3[e],
                       I.S[I.e_fn[e]], I.s0[e], I.eta[e * I.estride], I.qu[e * I.qstride]);
#endif
            }
        }
        team_sync(I);
        for (int e = tid; e < ne; e += nt)
            I.s3[e] = pdp_safe_log(1.0f - I.eta[e * I.estride], PDP_SCORER_EPS) * (0.0f + I.af[I.e_fn[e]]);
        team_sync(I);
        for (int v = tid; v < n; v += nt) {
            float ext = 0.0f, pos = 0.0f, neg = 0.0f, all = 0.0f;
            for (int k = I.v_ptr[v]; k < I.v_ptr[v + 1]; ++k) {
                const int e = I.v_edges[k];
                const int sg = I.sgn[e];
                const float f = I.s3[e];
                ext = ext + I.force[e * I.fstride];
                pos = pos + ((sg == 1) ? 1.0f : 0.0f) * f;
                neg = neg + ((sg == -1) ? 1.0f : 0.0f) * f;
                all = all + f;
            }
            const float sc = d_score_from_sums(pos, neg, all, ext, pi);
            if (check_score && sc != sc) {
                bad = 1;
#ifdef PDP_PHASE_PROF
                printf("[ghost] inst %d var %d: score NaN (pos %g neg %g all %g ext %g)\n", I.b, v, pos, neg, all, ext);
#endif
            }
        }
        return team_any(I, bad);
}

// HBM-resident form of the solver: the instance's arrays stay where the problem keeps them, one workgroup (TEAM = false) or a
// team of workgroups (TEAM = true, few big instances) walks them.
// LOCK: the whole (small) batch in lock step -- one workgroup per instance, every workgroup of the launch resident, and three batch-wide
// mailbox exchanges per iteration carry what the reference couples the instances through: the batch-global minima of the three
// sparse_max / sparse_argmax sites (pdp_decimate.py:127-160; an INACTIVE instance keeps contributing: its frozen survey maxima, zeros at
// the difference site, and -- it "converges" in every iteration -- its frozen |score| * active coefficients), a NaN anywhere, the
// replica-aware termination rule (trainer.py:157-160) and the global early exit.  Nothing is speculated, recorded or replayed.
struct LockNone {};
template <int NT, bool TEAM, bool LOCK = false>
__global__ void __launch_bounds__(NT) k_sp_solve(PView pv, SolveParams sp)
{
    static_assert(!(TEAM && LOCK), "lock-step batches give every instance one workgroup");
    __shared__ uint32_t lock_w[LOCK ? PDP_LOCK_MAX : 1];
    // (only per-wave reduction scratch in LDS: the instance's state is in HBM)
    __shared__ float redf[NT / PDP_WAVE];
    __shared__ int redi[NT / PDP_WAVE];
    __shared__ float red5[5 * (NT / PDP_WAVE)];

    if (sp.hbm_device_ctl) {
        if (sp.call->stop) return;                                           // every instance went inactive in an earlier chunk
        if (sp.hbm_replay && !sp.ctl->do_replay) return;                     // no NaN poisoned the batch in this chunk
    }
    int poison_from = sp.hbm_device_ctl ? (sp.hbm_replay ? sp.ctl->poison_from : (sp.call->poisoned_all ? 0 : 0x7fffffff)) : sp.poison_from;
    const bool exact = sp.exact != 0, rf = sp.rf != 0;
    const bool own_nan = exact || sp.isolate != 0 || LOCK;      // no batch-wide poison record, no replay (LOCK: the NaN is exchanged in the iteration)
    std::conditional_t<TEAM, TeamView, SView<int32_t>> I;
    int slot = (int)blockIdx.x;          // which of the launch's instances
    if constexpr (TEAM) {
        // the team is the long pole of a mixed batch and shares its CUs with the LDS-resident kernel's waves: let the scheduler prefer it
        __builtin_amdgcn_s_setprio(3);
        TeamLaunch tl; tl.size = sp.team_size; tl.count = sp.team_count; tl.slots = sp.team_slots; tl.no_xcd = sp.team_no_xcd; tl.ws = sp.team_ws;
        tl.spin_limit = sp.spin_limit;
        slot = team_begin(I, tl, redi);
        if (slot < 0) return;
    }
    const Inst G = load_inst(pv, __builtin_amdgcn_readfirstlane(sp.big_list ? sp.big_list[slot] : slot));
    const int tid = team_tid(I), nt = team_nt(I);
    constexpr int ROWL = TEAM ? 8 : (NT >= 1024 ? 4 : 1);      // lanes per variable row: the fewer threads an instance has, the more rows each must walk anyway
    const int rl = tid & (ROWL - 1);
    const int n = G.n, m = G.m, ne = G.e;
    float *gq = sp.q + 3 * (size_t)G.e0;
    float *gfs = sp.fs + 2 * (size_t)G.e0;
    const float L0h = pdp_safe_log(1.0f - sp.pi * 0.0f, PDP_SP_EPS), L1h = pdp_safe_log(1.0f - sp.pi * 1.0f, PDP_SP_EPS);

    I.b = G.b; I.n = n; I.m = m; I.e = ne;
    I.e_var = G.e_var; I.e_fn = G.e_fn; I.v_edges = G.v_edges; I.f_edges = G.f_edges; I.v_ptr = G.v_ptr; I.f_ptr = G.f_ptr;
    I.sgn = G.sgn; I.av = G.av; I.af = G.af; I.sol = G.sol; I.emask = G.emask;
    I.qu = gq; I.qstride = 3; I.eta = gfs; I.estride = 2; I.force = gfs + 1; I.fstride = 2;
    I.s0 = sp.ws_e[0] + G.e0; I.s1 = sp.ws_e[1] + G.e0; I.s2 = sp.ws_e[2] + G.e0; I.s3 = sp.ws_e[3] + G.e0;
    I.S = sp.ws_f + G.f0;
    I.P = sp.ws_v[0] + G.v0; I.N = sp.ws_v[1] + G.v0; I.xv1 = sp.ws_v[2] + G.v0; I.xv2 = sp.ws_v[3] + G.v0;
    I.score = sp.ws_v[4] + G.v0; I.coeff = sp.ws_v[5] + G.v0; I.assign = sp.ws_v[6] + G.v0;
    I.deg = sp.ws_vi[0] + G.v0; I.sdeg = sp.ws_vi[1] + G.v0;
    I.flag_v = reinterpret_cast<uint8_t *>(sp.ws_vi[2]) + G.v0;
    I.flag_f = sp.ws_fu[0] + G.f0; I.flag_f2 = sp.ws_fu[1] + G.f0;

    SimplifyScratch ss;
    ss.assign = I.assign; ss.deg = I.deg; ss.sdeg = I.sdeg; ss.flag_v = I.flag_v; ss.flag_f = I.flag_f; ss.flag_f2 = I.flag_f2; ss.red = redi;
    // Per-edge record.  The variable rows reach their edges through the by-variable CSR -- random 4-byte gathers, a 64-byte line each: eight of
    // them per edge and sweep were what a huge instance waited for (n = 1 000 000: rows 0.88 + maxima 0.73 of 2.16 ms).  Everything a row needs
    // from an edge sits in one 16-byte record, written contiguously by the per-edge passes: one gather in P2, one in P4.
    float4 *const R = reinterpret_cast<float4 *>(sp.ws_r) + G.e0;
    for (int e = tid; e < ne; e += nt)
        R[e] = make_float4(gfs[2 * e], 1.0f, 0.0f, (float)G.sgn[e] * ((!sp.has_edge_mask || G.emask[e] == 1.0f) ? 1.0f : 2.0f));
    team_sync(I);

    int active = sp.amask[G.b] ? 1 : 0;
    int has_prev = sp.has_prev;
    int prev_from_global = sp.has_prev;       // first iteration compares with the decimator's stored survey
    int use_em = sp.has_edge_mask;            // sat_problem._edge_mask is not None
    float cnt = sp.counters[G.b];
    int iters = 0, did_prop = 0;
    int nsat = -1;                            // cached CNF result (solution only changes on decimation)
    int violation = 0;
    int rf_last_flip = 0;                     // Reinforce: the force was renewed after the last sweep (s2 holds the one that sweep read)
    const bool other_rows = n < pv.V;

    int abort_next = 0;
    // ---- LOCK: the batch as a team (mailboxes only: no instance data crosses workgroups), frozen contributions of an inactive instance
    Teamed<LockNone> BT;
    float fz_mn0 = PDP_INF, fz_mnc = PDP_INF;
    int any_active = 1;
    auto frozen = [&]() {
        float mn = PDP_INF;
        for (int v = tid; v < n; v += nt) {
            float num = 0.0f, den = 0.0f;
            for (int k = I.v_ptr[v]; k < I.v_ptr[v + 1]; ++k) {
                const float et = I.eta[I.v_edges[k] * I.estride];
                const float c1 = pdp_safe_exp(30.0f * et);
                num = num + et * c1; den = den + c1;
            }
            mn = OpMinLess()(mn, (num / pdp_max(den, 1.0f)) * I.av[v]);
        }
        fz_mn0 = block_reduce(mn, OpMinLess(), PDP_INF, redf);
        for (int e = tid; e < ne; e += nt)
            I.s3[e] = pdp_safe_log(1.0f - I.eta[e * I.estride], PDP_SCORER_EPS) * (0.0f + I.af[I.e_fn[e]]);
        __syncthreads();
        mn = PDP_INF;
        for (int v = tid; v < n; v += nt) {
            float ext = 0.0f, pos = 0.0f, neg = 0.0f, all = 0.0f;
            for (int k = I.v_ptr[v]; k < I.v_ptr[v + 1]; ++k) {
                const int e = I.v_edges[k];
                const int sg = I.sgn[e];
                const float f = I.s3[e];
                ext = ext + I.force[e * I.fstride];
                pos = pos + ((sg == 1) ? 1.0f : 0.0f) * f;
                neg = neg + ((sg == -1) ? 1.0f : 0.0f) * f;
                all = all + f;
            }
            mn = OpMinLess()(mn, (pdp_abs(d_score_from_sums(pos, neg, all, ext, sp.pi)) * I.av[v]) * 1.0f);
        }
        fz_mnc = block_reduce(mn, OpMinLess(), PDP_INF, redf);
    };
    // An INACTIVE instance is not swept here, but the reference still computes its next messages in every iteration and blends them away
    // with the active mask: mask * new + (1 - mask) * old (pdp_propagate.py:219-221, pdp_decimate.py:230).  0 * new is NaN when `new` is not
    // finite (0/0 in the normalisation of a saturated instance), so such an instance's messages DO change -- and poison the batch.  The
    // frozen state is a fixed point of that computation unless its first "ghost" sweep yields a non-finite value, so the instance checks
    // exactly that once, when it goes inactive with iterations still to come: the would-be next sweep and the scorer (whose NaN reaches the
    // force column of Reinforce and the coefficient sum of the sequential decimator).  A hit fails the call over to the strict step-wise
    // loop, which evaluates the blends literally.
    // (scorer: a NaN score reaches the sequential decimator's coefficient sum; Reinforce takes torch.sign of it, which is 0 for a NaN)
    auto ghost_bad = [&]() -> int { return d_ghost_bad(I, sp.pi, L0h, L1h, use_em != 0, !rf); };
    if constexpr (LOCK) {
        BT.rank = (int)blockIdx.x; BT.size = sp.lock_size ? sp.lock_size : (int)gridDim.x; BT.epoch = 0; BT.same_xcd = 2; BT.bar = sp.team_ws; BT.box = BT.bar + 32;
        BT.failed = 0u; BT.spin_limit = sp.spin_limit ? sp.spin_limit : PDP_SPIN_LIMIT_DEFAULT;
        if (!active) frozen();
    }
#ifdef PDP_PHASE_PROF
    long long tacc[8] = {0, 0, 0, 0, 0, 0, 0, 0}; long long tlast = wall_clock64();
#define TP(i) { const long long now_ = wall_clock64(); tacc[i] += now_ - tlast; tlast = now_; }
#else
#define TP(i)
#endif
    for (int t = 0; t < sp.T; ++t) {
        if (!LOCK && !active) break;
        if (abort_next) break;                // pass 1 only: some instance poisons the batch before t, this pass will be replayed anyway
        bool poisoned = t >= poison_from;
        iters = t + 1;
        const int was_active = active;
        int nan_seen = 0, z1 = 0, z2 = 0;
        IterRed red; red.a = -PDP_INF; red.b = -PDP_INF; red.mna = PDP_INF; red.mnb = PDP_INF;
        if (!LOCK || active) {
        // ---- P1 + P2: per-edge logs (pdp_propagate.py:166-169,185-188) and their per-clause / per-variable sums in ascending
        // edge order; every edge sits in exactly one clause row and one variable row, which computes and leaves its log for P3
        // (memory latency is what this kernel waits for: rows are walked in batches whose loads are all in flight together, and
        // the sums then take the batch's values in entry order -- the reference's summation order)
        for (int c = tid; c < m; c += nt) {
            const int beg = I.f_ptr[c], end = I.f_ptr[c + 1];
            float acc = 0.0f;
            for (int k0 = beg; k0 < end; k0 += 4) {
                int ee[4]; float xx[4];
#pragma unroll
                for (int j = 0; j < 4; ++j) ee[j] = I.f_edges[(k0 + j < end) ? k0 + j : end - 1];
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    xx[j] = pdp_safe_log(I.qu[ee[j] * I.qstride], PDP_SP_EPS);
                    if (use_em) xx[j] = xx[j] * I.emask[ee[j]];
                }
#pragma unroll
                for (int j = 0; j < 4; ++j) if (k0 + j < end) { I.s0[ee[j]] = xx[j]; acc = acc + xx[j]; }
            }
            I.S[c] = acc;
        }
        // a variable row belongs to ROWL adjacent lanes: each gathers one entry, every lane of the group adds them in entry order
        for (int v = tid / ROWL; v < n; v += nt / ROWL) {
            const int beg = I.v_ptr[v], end = I.v_ptr[v + 1];
            float P = 0.0f, N = 0.0f;
            for (int k0 = beg; k0 < end; k0 += ROWL) {
                const int k = k0 + rl;
                float y = 0.0f; int sg = 0;
                if (k < end) {
                    const float4 r = R[I.v_edges[k]];
                    y = pdp_safe_log(1.0f - r.x, PDP_SP_EPS);
                    if (use_em) y = y * ((pdp_abs(r.w) == 1.0f) ? 1.0f : 0.0f);
                    sg = (r.w > 0.0f) ? 1 : -1;
                }
                const int cntk = end - k0;
#pragma unroll
                for (int j = 0; j < ROWL; ++j) {
                    const float yj = __shfl(y, j, ROWL);
                    const int sj = __shfl(sg, j, ROWL);
                    if (j < cntk) {
                        P = P + ((sj == 1) ? 1.0f : 0.0f) * yj;
                        N = N + ((sj == -1) ? 1.0f : 0.0f) * yj;
                    }
                }
            }
            if (rl == 0) { I.P[v] = P; I.N[v] = N; }
        }
        TP(0)
        team_sync(I);
        TP(1)
        // ---- P3: new surveys + new q_u, and the decimator's per-edge terms (pdp_decimate.py:128-141)
        for (int e = tid; e < ne; e += nt) {
            const int v = I.e_var[e], c = I.e_fn[e];
            const float s = (float)I.sgn[e];
            const float eta_old = I.eta[e * I.estride];
            const float agg = (0.0f + I.S[c]) - I.s0[e];
            const float eta_new = 1.0f * pdp_safe_exp(agg) + (1.0f - 1.0f) * eta_old;
            const float force = I.force[e * I.fstride];
            float y = pdp_safe_log(1.0f - eta_old, PDP_SP_EPS);          // the edge's term of its variable row (P2 computed the same value from the record)
            if (use_em) y = y * I.emask[e];
            I.s1[e] = y;                                                  // (the write-back rebuilds q_s / q_dc of the last sweep from it)
            const SpOut o = d_sp_edge(s, I.P[v], I.N[v], y, force, L0h, L1h);
            const float qu_old = I.qu[e * I.qstride];
            const float qu_new = 1.0f * o.qu + (1.0f - 1.0f) * qu_old;
            // only a NaN SURVEY poisons the batch-global reductions of this iteration; a NaN in q (0/0) reaches the
            // surveys one iteration later (x = log(max(NaN, eps)) = NaN)
            if (eta_new != eta_new) nan_seen = 1;
            float d = 0.0f;
            if (has_prev) {
                const float pe = prev_from_global ? sp.prev[G.e0 + e] : eta_old;
                d = pdp_abs(pe - eta_new);
                if (use_em) d = d * I.emask[e];
            }
            I.qu[e * I.qstride] = qu_new;
            I.eta[e * I.estride] = eta_new;
            // (the smooth-max weight of the survey, exp(30 eta), is taken by P4 from the record's survey)
            float *rec = reinterpret_cast<float *>(&R[e]);
            rec[0] = eta_new; rec[1] = pdp_safe_exp(30.0f * d); rec[2] = d;      // smooth-max weight of the difference, the difference
        }
        did_prop = 1;
        TP(2)
        team_sync(I);
        TP(3)
        // ---- P4 + P5: per-variable smooth maxima (util.py:282-286) times the active flag, and their per-instance maxima with
        // the reference's (x - min + 1) rounding (util.sparse_max, util.py:267-275), speculating min == 0
        for (int v = tid / ROWL; v < n; v += nt / ROWL) {
            const int beg = I.v_ptr[v], end = I.v_ptr[v + 1];
            float num1 = 0.0f, den1 = 0.0f, num2 = 0.0f, den2 = 0.0f;
            for (int k0 = beg; k0 < end; k0 += ROWL) {
                const int k = k0 + rl;
                float c1 = 0.0f, p1 = 0.0f, c2 = 0.0f, p2 = 0.0f;
                if (k < end) {
                    const float4 r = R[I.v_edges[k]];
                    c1 = pdp_safe_exp(30.0f * r.x); p1 = r.x * c1;
                    if (has_prev) { c2 = r.y; p2 = r.z * c2; }
                }
                const int cntk = end - k0;
#pragma unroll
                for (int j = 0; j < ROWL; ++j) {
                    const float c1j = __shfl(c1, j, ROWL), p1j = __shfl(p1, j, ROWL);
                    if (j < cntk) { num1 = num1 + p1j; den1 = den1 + c1j; }
                    if (has_prev) {
                        const float c2j = __shfl(c2, j, ROWL), p2j = __shfl(p2, j, ROWL);
                        if (j < cntk) { num2 = num2 + p2j; den2 = den2 + c2j; }
                    }
                }
            }
            const float a = I.av[v];
            const float r1 = (num1 / pdp_max(den1, 1.0f)) * a;
            red.a = pdp_max(red.a, r1); red.mna = OpMinLess()(red.mna, r1);
            if (r1 == 0.0f) z1 = 1;
            if (r1 != r1) nan_seen = 1;
            if (has_prev) {
                const float r2 = (num2 / pdp_max(den2, 1.0f)) * a;
                red.b = pdp_max(red.b, r2); red.mnb = OpMinLess()(red.mnb, r2);
                if (r2 == 0.0f) z2 = 1;
                if (r2 != r2) nan_seen = 1;
            }
        }
        TP(4)
        } else {
            // LOCK, inactive instance: its frozen messages still enter the batch-global minima of this iteration
            if (n > 0) { red.mna = fz_mn0; if (has_prev) red.mnb = 0.0f; }
        }
        // the same barrier carries pass 1's look at the batch's first NaN iteration for the next trip of the loop
        const int nan_before_next = (!own_nan && poison_from == 0x7fffffff && tid == 0 &&
                                     __hip_atomic_load(sp.nan_iter, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < (uint32_t)(t + 1)) ? 1 : 0;
        red.bits = (z1 ? 1 : 0) | (z2 ? 2 : 0) | (nan_seen ? 4 : 0) | (nan_before_next ? 8 : 0);
        red = team_iter_reduce(I, red, exact || LOCK, red5);
        TP(5)
        z1 = red.bits & 1; z2 = red.bits & 2; nan_seen = red.bits & 4; abort_next = red.bits & 8;
        // util.sparse_max (util.py:267-275): max_v((x_v - min) + 1), then + min - 1.  Rounding is monotone, so the maximum of the
        // shifted values is the shifted maximum and the raw maximum is all the reduction has to carry.  min: the batch-global one --
        // speculated to be 0, or (single-instance batch) the instance's own.
        float gm1 = exact ? red.mna : 0.0f, gm2 = exact ? red.mnb : 0.0f;
        if constexpr (LOCK) {
            IterRed ex; ex.a = -PDP_INF; ex.b = -PDP_INF; ex.mna = red.mna; ex.mnb = red.mnb; ex.bits = red.bits & 4;
            ex = team_iter_reduce(BT, ex, true, red5);
            gm1 = ex.mna; gm2 = ex.mnb;
            if ((ex.bits & 4) && !poisoned) { poison_from = t; poisoned = true; }       // a NaN survey anywhere poisons the whole batch from this iteration on
        }
        float g = (red.a - gm1) + 1.0f, dmax = 0.0f;
        if (other_rows) g = pdp_max(g, 0.0f);
        g = (g + gm1) - 1.0f;
        if (has_prev) {
            dmax = (red.b - gm2) + 1.0f;
            if (other_rows) dmax = pdp_max(dmax, 0.0f);
            dmax = (dmax + gm2) - 1.0f;
        }
        // single-instance batch: its own NaN survey poisons it from this iteration on, before any decision of the iteration is made
        if (exact && nan_seen && !poisoned) { poison_from = t; poisoned = true; }
        // A NaN survey (0/0 in pdp_propagate.py:215-216) makes every batch-global min/max of the reference NaN from
        // this iteration on.  Pass 1 records the first such iteration, pass 2 replays with it.
        if (nan_seen && !poisoned && sp.isolate) {
            // isolated instance: comparisons with the NaN maxima are false below and `nan_seen` keeps the decimation off, that is all
        } else if (nan_seen && !poisoned) {
            if (tid == 0) atomicMin(sp.nan_iter, (uint32_t)t);
            if (poison_from != 0x7fffffff) violation = 1;      // pass 2 must not find an earlier NaN
            else abort_next = 1;
        }
        int conv = 0, rf_changed = 0;
        if (LOCK && !active) {
            // (an inactive instance takes no decision)
        } else if (rf) {
            // active_mask[sum_diff <= 0.01] = 0 (pdp_decimate.py:205-215; no survey gate, no counters); under the poison the batch-wide maximum is NaN
            if (!poisoned && has_prev && dmax <= sp.tol) active = 0;
        } else if (!poisoned) {
            if (g <= 1e-10f) active = 0;                      // trivial surveys: leave the instance to Walk-SAT
            if (has_prev) {
                if (dmax < sp.tol) cnt = 0.0f;
                conv = (dmax < sp.tol) ? 1 : 0;
                if (cnt >= sp.t_max) { conv = 1; cnt = 0.0f; }
            }
        } else if (has_prev) {
            // all maxima are NaN: comparisons are False, only the counter overflow still "converges" an instance
            if (cnt >= sp.t_max) { conv = 1; cnt = 0.0f; }
        }
        uint32_t used = (rf ? 0u : 1u) | (has_prev ? 2u : 0u);
        uint32_t zero = ((!rf && z1) ? 1u : 0u) | ((has_prev && z2) ? 2u : 0u);
        // coeff = |score| * active * converged: every variable of a non-converged instance is an exact 0 of site 2
        if (!rf && has_prev && !conv && n > 0) zero |= 4u;
        if (rf && (!LOCK || was_active)) {
            // the shared coin of this iteration (pdp_decimate.py:218); instances that are still active after the gate renew their force.
            // Predictor + _update_solution run in every iteration of the reference; their result only changes with the force.
            const int flip = (sp.coins[sp.chunk_start + t] < sp.dprob) && active;
            rf_last_flip = flip;
            if (flip || nsat < 0) {
                if (hbm_reinforce_step(I, gfs, sp.pi, flip) && !nan_seen) violation = 1;    // a NaN score without a NaN survey: not expected
                rf_changed = 1;
            }
        }
        // ---- P6: decimation (pdp_decimate.py:152-171)
        int decimated = 0;
        const bool will = !rf && has_prev && conv && !poisoned && !nan_seen && (!LOCK || was_active);    // (an instance the gate just closed still enters site 2)
        int z3 = 0, anynz = 0, cn = 0;
        float gm3 = 0.0f;
        if (will) {
            // scorer (pdp_predict.py:155-192)
            for (int e = tid; e < ne; e += nt)
                I.s3[e] = pdp_safe_log(1.0f - I.eta[e * I.estride], PDP_SCORER_EPS) * (0.0f + I.af[I.e_fn[e]]);
            team_sync(I);
            for (int v = tid; v < n; v += nt) {
                float ext = 0.0f, pos = 0.0f, neg = 0.0f, all = 0.0f;
                for (int k = I.v_ptr[v]; k < I.v_ptr[v + 1]; ++k) {
                    const int e = I.v_edges[k];
                    const int sg = I.sgn[e];
                    const float f = I.s3[e];
                    ext = ext + I.force[e * I.fstride];
                    pos = pos + ((sg == 1) ? 1.0f : 0.0f) * f;
                    neg = neg + ((sg == -1) ? 1.0f : 0.0f) * f;
                    all = all + f;
                }
                const float sc = d_score_from_sums(pos, neg, all, ext, sp.pi);
                const float co = (pdp_abs(sc) * I.av[v]) * 1.0f;
                I.score[v] = sc; I.coeff[v] = co;
                if (co == 0.0f) z3 = 1;
                if (co != 0.0f) anynz = 1;
                if (co != co) cn = 1;
            }
            {
                const int bits = team_reduce(I, (z3 ? 1 : 0) | (anynz ? 2 : 0) | (cn ? 4 : 0), OpOrI(), 0, redi);     // (its barrier also publishes score / coeff)
                z3 = bits & 1; anynz = bits & 2; cn = bits & 4;
            }
            if (cn) violation = 1;                                // cannot happen without a NaN survey
            used |= 4u; if (z3) zero |= 4u;
            if (exact || LOCK) {
                float mn = PDP_INF;
                for (int v = tid; v < n; v += nt) mn = OpMinLess()(mn, I.coeff[v]);
                gm3 = team_reduce(I, mn, OpMinLess(), PDP_INF, redf);
            }
        }
        if constexpr (LOCK) {
            // site 2 over the batch: a converged instance's smallest coefficient, 0 for an active one that did not converge (all its
            // coefficients are 0), the frozen one of an inactive instance
            if (has_prev && !rf) gm3 = team_reduce(BT, will ? gm3 : (was_active ? (n > 0 ? 0.0f : PDP_INF) : fz_mnc), OpMinLess(), PDP_INF, redf);
        }
        if (will) {
            const int li = d_instance_argmax(I, I.coeff, gm3, redf, redi);
            if (active && anynz && !cn && li >= 0) {
                for (int v = tid; v < n; v += nt) I.assign[v] = 0.0f;
                team_sync(I);
                if (tid == 0) I.assign[li] = pdp_sign(I.score[li]);
                team_sync(I);
                d_set_variable_core(I, ss);
                d_simplify(I, ss, pv.is_sat + G.b);
                decimated = 1;
            }
        }
        if (has_prev && (!LOCK || was_active)) cnt = cnt + 1.0f;
        if (tid == 0 && !poisoned && !own_nan) { atomicOr(&sp.spec_used[t], used); if (zero) atomicOr(&sp.spec_zero[t], zero); }
        // ---- P7: edge mask refresh (solver.py:370-371); values only change after a decimation
        if (decimated || !use_em) {
            for (int e = tid; e < ne; e += nt) {
                const float a = 0.0f + I.av[I.e_var[e]];
                const float b = 0.0f + I.af[I.e_fn[e]];
                I.emask[e] = a * b;
                reinterpret_cast<float *>(&R[e])[3] = (float)I.sgn[e] * ((a * b == 1.0f) ? 1.0f : 2.0f);
            }
            use_em = 1;
            team_sync(I);
        }
        // ---- P8: prediction = solution; termination check (trainer.py:150-162, util.py:226-236)
        if (sp.check_termination) {
            if (decimated || rf_changed || nsat < 0) nsat = d_cnf_sat_count(I, I.sol, redi);
            if (active && nsat == m) active = 0;
        }
        if constexpr (LOCK) {
            // termination across the batch: replicas of a solved instance stop together (trainer.py:157-160), everybody stops when nobody is left
            uint32_t *box = team_box(BT);
            if (threadIdx.x == 0) box_put(&box[PDP_BOX_WORDS * BT.rank], ((sp.check_termination && nsat == m) ? 1u : 0u) | (active ? 2u : 0u));
            team_sync(BT);
            for (int r = (int)threadIdx.x; r < BT.size; r += (int)blockDim.x) lock_w[r] = box_get(&box[PDP_BOX_WORDS * r]);
            __syncthreads();
            auto group_solved = [&](int b) { int s_ = 0; for (int r = 0; r < pv.R; ++r) s_ |= (int)(lock_w[b % pv.B0 + r * pv.B0] & 1u); return s_; };
            if (pv.R > 1 && sp.check_termination && active && group_solved(G.b)) active = 0;
            int still = 0;
            for (int r = (int)threadIdx.x; r < BT.size; r += (int)blockDim.x) still |= ((lock_w[r] & 2u) && !(pv.R > 1 && sp.check_termination && group_solved(r))) ? 1 : 0;
            any_active = __syncthreads_or(still);
            if (was_active && !active) { frozen(); if (t + 1 < sp.T && any_active && !sp.isolate && ghost_bad()) violation = 1; }
        }
        has_prev = 1; prev_from_global = 0;
        TP(6)
        if (LOCK && !any_active) break;
    }
#ifdef PDP_PHASE_PROF
    if (tid == 0 && TEAM) printf("[team %d x %d] iters %d: rows %lld sync %lld edges %lld sync %lld maxima %lld reduce %lld rest %lld (x10 ns)\n", I.b, nt, iters, tacc[0], tacc[1], tacc[2], tacc[3], tacc[4], tacc[5], tacc[6]);
#endif
#undef TP

    if (!LOCK && !active && did_prop && !sp.isolate && !(sp.final_chunk && iters >= sp.T)) { if (ghost_bad()) violation = 1; }
    // ---- write back -------------------------------------------------------------------------------------------
    if (did_prop) {
        // q_s / q_dc of the last sweep are recomputed from the per-variable sums that are still resident
        for (int e = tid; e < ne; e += nt) {
            const float frc = (rf && rf_last_flip) ? I.s2[e] : I.force[e * I.fstride];      // the force the last sweep read
            const SpOut o = d_sp_edge((float)I.sgn[e], I.P[I.e_var[e]], I.N[I.e_var[e]], I.s1[e], frc, L0h, L1h);
            // (1 - mask) * old keeps a NaN forever; the three columns of q turn NaN together, so q_u carries the stickiness
            const float sticky = I.qu[e * I.qstride];
            gq[3 * e + 1] = 1.0f * o.qs + (1.0f - 1.0f) * sticky;
            gq[3 * e + 2] = 1.0f * o.dc + (1.0f - 1.0f) * sticky;
            sp.prev[G.e0 + e] = I.eta[e * I.estride];
        }
    }
    // an instance with a de-activated variable contributes an exact 0 to every batch-global min from now on
    int any_inactive = 0;
    for (int v = tid; v < n; v += nt) any_inactive |= (I.av[v] == 0.0f) ? 1 : 0;
    any_inactive = team_any(I, any_inactive);
    // a team barrier (or the lock-step batch's) gave up: what this launch computed is void -- reported like a failed speculation, the host
    // restores the call-entry state and fails over to the strict step-wise loop
    if constexpr (TEAM) { if (I.failed) violation = 1; }
    if constexpr (LOCK) { if (BT.failed) violation = 1; }
    if (tid == 0) {
        if (any_inactive) atomicMin(sp.w_perm_zero, (uint32_t)iters);
        sp.amask[G.b] = (uint8_t)active;
        sp.counters[G.b] = cnt;
        atomicMax(sp.w_iters_run, (uint32_t)iters);
        if (violation) atomicOr(sp.w_violation, 1u);
    }
}


// =====================================================================================================================
// v2 LDS-resident kernel.  Canonical slot order is VARIABLE-MAJOR: slot p holds the p-th entry of the instance's
// by-variable CSR, so every per-variable sum reads a contiguous LDS range in the reference's summation order, and
// the few per-clause sums (k entries) go through e2p.  Per slot: 5 floats (q_u, two survey buffers, two scratch) and
// 6 bytes of packed topology -> ~80 KB for n=200/m=840, i.e. TWO workgroups per CU.
// Requires clause-major edge order (f_edges == identity), instances < 16384 variables / clauses.
// =====================================================================================================================
struct LView {   // what the shared simplification routines see: "edge id" == slot
    int b, n, m, e;
    int nt;      // workgroup size (team_nt: the out-of-line routines must not read blockDim from the dispatch packet)
    int *red;    // >= PDP_RED_SMALL words of LDS scratch for team_any
    struct EVar { const uint16_t *pv; int mask; __device__ __forceinline__ int operator[](int p) const { return pv[p] & mask; } } e_var;   // mask 0x3fff; 0x1fff when bits 13-14 hold the Reinforce force
    struct EFn { const uint16_t *pc; __device__ __forceinline__ int operator[](int p) const { return pc[p] & 0x3fff; } } e_fn;
    struct Sgn { const uint16_t *pv; __device__ __forceinline__ int operator[](int p) const { return (pv[p] & 0x8000) ? -1 : 1; } } sgn;
    struct Iden { __device__ __forceinline__ int operator[](int k) const { return k; } } v_edges;
    const uint16_t *f_edges;     // e2p
    const uint16_t *v_ptr, *f_ptr;
    float *av, *af, *sol;
};

#define PDP_RED_SMALL 16    /* per-wave reduction slots of the LDS solver (<= 16 waves per workgroup) */
/* Reinforce triple: the external force of a slot is one of {0, +1, -1, NaN} (initial state 0, then torch.sign of a score) and lives in bits
 * 13-14 of the slot's variable word, so the instance image stays at five floats per slot and two workgroups still share a CU; the variable
 * id then has 13 bits (instances of < 8192 variables) */
#define PV_FRC_SHIFT 13
#define PV_FRC_MASK 0x6000u
#define PV_VMASK_RF 0x1fff
__host__ __device__ __forceinline__ uint32_t frc_enc(float f) { return (f != f) ? 3u : ((f == 1.0f) ? 1u : ((f == -1.0f) ? 2u : 0u)); }
__device__ __forceinline__ float frc_dec(uint32_t c) { return __uint_as_float((c == 0u) ? 0u : ((c == 1u) ? 0x3f800000u : ((c == 2u) ? 0xbf800000u : 0x7fc00000u))); }
__device__ __forceinline__ float frc_of(uint16_t pw) { return frc_dec(((uint32_t)pw & PV_FRC_MASK) >> PV_FRC_SHIFT); }
#define PC_EM 0x8000u      /* current edge mask bit */
#define PC_EM_USED 0x4000u /* edge mask bit the last propagate used (needed to rebuild q_s / q_dc at exit) */

static size_t lds2_bytes_for(int n, int m, int e)
{
    auto a16 = [](size_t x) { return (x + 15) & ~(size_t)15; };
    size_t s = 0;
    s += 5 * a16((size_t)e * 4);                               // QU, EA, EB, X, Y (an external force is a 2-bit code in the slot word)
    s += 3 * a16((size_t)e * 2);                               // pv, pc, e2p
    s += a16((size_t)(n + 1) * 2) + a16((size_t)(m + 1) * 2);  // v_ptr, f_ptr
    s += a16((size_t)(m + 8) * 4) + a16((size_t)m * 4);        // S (aliases flag_f/flag_f2), af
    s += 7 * a16((size_t)n * 4);                               // av, sol, P, N, xv1(deg), xv2(sdeg,score), coeff(assign)
    s += a16((size_t)n);                                       // flag_v
    s += a16((size_t)n * 2);                                   // vord
    return s;
}

// every LDS array of the v2 kernel, derived from (base, n, m, e): cold code re-derives them instead of keeping ~25
// pointers alive across the hot loop (they would spill out of the SGPR file)
struct LdsArrays {
    float *QU, *EA, *EB, *X, *Y;
    uint16_t *pvv, *pcc, *e2p, *v_ptr, *f_ptr;
    float *S, *af, *av, *sol, *Pv, *Nv, *xv1, *xv2, *coeff;
    uint8_t *flag_v;
    uint16_t *vord;      // variables in order of descending degree: a wave of 64 consecutive entries runs loops of similar length
};
__device__ __forceinline__ LdsArrays carve_all(unsigned char *cp, int n, int m, int ne)
{
    LdsArrays L;
    L.QU = carve<float>(cp, ne); L.EA = carve<float>(cp, ne); L.EB = carve<float>(cp, ne); L.X = carve<float>(cp, ne); L.Y = carve<float>(cp, ne);
    L.pvv = carve<uint16_t>(cp, ne); L.pcc = carve<uint16_t>(cp, ne); L.e2p = carve<uint16_t>(cp, ne);
    L.v_ptr = carve<uint16_t>(cp, n + 1); L.f_ptr = carve<uint16_t>(cp, m + 1);
    L.S = carve<float>(cp, m + 8); L.af = carve<float>(cp, m);
    L.av = carve<float>(cp, n); L.sol = carve<float>(cp, n); L.Pv = carve<float>(cp, n); L.Nv = carve<float>(cp, n);
    L.xv1 = carve<float>(cp, n); L.xv2 = carve<float>(cp, n); L.coeff = carve<float>(cp, n);
    L.flag_v = carve<uint8_t>(cp, n);
    L.vord = carve<uint16_t>(cp, n);
    return L;
}
__device__ __forceinline__ LView make_lview(const LdsArrays &L, int b, int n, int m, int ne, int nt, int *red, int vmask = 0x3fff)
{
    LView I;
    I.b = b; I.n = n; I.m = m; I.e = ne; I.nt = nt; I.red = red;
    I.e_var.pv = L.pvv; I.e_var.mask = vmask; I.e_fn.pc = L.pcc; I.sgn.pv = L.pvv; I.f_edges = L.e2p; I.v_ptr = L.v_ptr; I.f_ptr = L.f_ptr;
    I.av = L.av; I.af = L.af; I.sol = L.sol;
    return I;
}
#define UNI(x) __builtin_amdgcn_readfirstlane(x)
#ifdef PDP_PHASE_PROF
__device__ unsigned long long g_phase_cycles[40];
// per-phase sums stay in registers and are flushed once per launch: one atomic per phase and iteration from 5000 workgroups
// onto the same 16 words more than doubled the kernel time
#define PROF_DECL unsigned long long _t0 = __builtin_readcyclecounter(), _t1; uint32_t _acc[20] = {0};
#define PROF_MARK(i) do { _t1 = __builtin_readcyclecounter(); _acc[i] += (uint32_t)(_t1 - _t0); _t0 = _t1; } while (0)
#define PROF_FLUSH() do { if (threadIdx.x == 0) { _Pragma("unroll") for (int _i = 0; _i < 20; ++_i) if (_acc[_i]) atomicAdd(&g_phase_cycles[_i < 13 ? _i : _i + 11], (unsigned long long)_acc[_i]); } } while (0)
extern "C" int pdp_debug_phase_cycles(unsigned long long *out_host, int reset)
{
    if (hipMemcpyFromSymbol(out_host, HIP_SYMBOL(g_phase_cycles), sizeof(unsigned long long) * 40) != hipSuccess) return 1;
    if (reset) { unsigned long long z[40] = {0}; if (hipMemcpyToSymbol(HIP_SYMBOL(g_phase_cycles), z, sizeof(z)) != hipSuccess) return 1; }
    return 0;
}
#define PROF_COUNT(i) do { _acc[i] += 1u; } while (0)
#define PROF_SKIP(bit) (sp.debug_skip & (bit))
#define DEC_PROF_DECL unsigned long long _d0 = __builtin_readcyclecounter(), _d1;
#define DEC_PROF_MARK(i) do { _d1 = __builtin_readcyclecounter(); if (threadIdx.x == 0) atomicAdd(&g_phase_cycles[i], _d1 - _d0); _d0 = _d1; } while (0)
#else
#define PROF_DECL
#define PROF_MARK(i)
#define PROF_COUNT(i)
#define PROF_FLUSH()
#define PROF_SKIP(bit) false
#define DEC_PROF_DECL
#define DEC_PROF_MARK(i)
#endif
// Four independent evaluations side by side.  A dependent chain of VALU ops issues one instruction per ~4.3 cycles on
// gfx950, two or more independent chains in the same wave reach ~2.2 (tools/micro/pk_rate.hip), so the transcendental
// polynomials of one slot (four exps) or two slots (four logs) are evaluated as 4-vectors: every step is the same IEEE
// operation per element as the scalar pdp_expf_fin_le30 / pdp_safe_log_fin of include/pdp_math.h.
typedef float f4v __attribute__((ext_vector_type(4)));
typedef int i4v __attribute__((ext_vector_type(4)));
typedef unsigned u4v __attribute__((ext_vector_type(4)));
typedef float f2v __attribute__((ext_vector_type(2)));
typedef int i2v __attribute__((ext_vector_type(2)));
typedef unsigned u2v __attribute__((ext_vector_type(2)));
__device__ __forceinline__ f4v vfma(f4v a, f4v b, f4v c) { return __builtin_elementwise_fma(a, b, c); }

#ifdef PDP_FAST_MATH
// the opt-in fast build (include/pdp_math.h, PDP_FAST_MATH): the same split as pdp_hw_expf / pdp_hw_logf, four wide.  The arguments of
// this kernel are finite sums of at most 16 383 clamped logs (> -1.6e6) or NaN: the low part of the product stays below 1/4, nothing to clamp.
__device__ __forceinline__ f4v exp4_fin_le30(f4v x)
{
    const f4v t = x * 1.44269502162933349609375f;
    const f4v n = __builtin_elementwise_rint(t);
    f4v lo = vfma(x, (f4v)(1.44269502162933349609375f), -t);
    lo = vfma(x, (f4v)(1.92596299112661746e-8f), lo);
    const f4v f = (t - n) + lo;
    const i4v ni = __builtin_convertvector(n, i4v);
    f4v res;
    res.x = __builtin_ldexpf(__builtin_amdgcn_exp2f(f.x), ni.x); res.y = __builtin_ldexpf(__builtin_amdgcn_exp2f(f.y), ni.y);
    res.z = __builtin_ldexpf(__builtin_amdgcn_exp2f(f.z), ni.z); res.w = __builtin_ldexpf(__builtin_amdgcn_exp2f(f.w), ni.w);
    return res;
}
__device__ __forceinline__ f4v log4_fin(f4v x, float eps)
{
    const f4v xm = __builtin_bit_cast(f4v, __builtin_elementwise_max(__builtin_bit_cast(i4v, x), __builtin_bit_cast(i4v, (f4v)(eps))));
    f4v l; i4v e;
    l.x = __builtin_amdgcn_logf(__builtin_amdgcn_frexp_mantf(xm.x)); l.y = __builtin_amdgcn_logf(__builtin_amdgcn_frexp_mantf(xm.y));
    l.z = __builtin_amdgcn_logf(__builtin_amdgcn_frexp_mantf(xm.z)); l.w = __builtin_amdgcn_logf(__builtin_amdgcn_frexp_mantf(xm.w));
    e.x = __builtin_amdgcn_frexp_expf(xm.x); e.y = __builtin_amdgcn_frexp_expf(xm.y); e.z = __builtin_amdgcn_frexp_expf(xm.z); e.w = __builtin_amdgcn_frexp_expf(xm.w);
    const f4v r = (__builtin_convertvector(e, f4v) + l) * 0.693147180559945309f;
    return vfma((f4v)(0.0f), x, r);          // a NaN whose sign bit is set went to eps in the integer maximum: re-injected
}
__device__ __forceinline__ f4v exp4_sum(f4v x) { return exp4_fin_le30(x); }
__device__ __forceinline__ float exp1_sum(float x)          // one element of exp4_fin_le30 as a scalar chain
{
    const float t = x * 1.44269502162933349609375f;
    const float n = __builtin_rintf(t);
    float lo = __builtin_fmaf(x, 1.44269502162933349609375f, -t);
    lo = __builtin_fmaf(x, 1.92596299112661746e-8f, lo);
    const float f = (t - n) + lo;
    return __builtin_ldexpf(__builtin_amdgcn_exp2f(f), (int)n);
}
__device__ __forceinline__ f2v log2_fin(f2v x, float eps)
{
    const f2v xm = __builtin_bit_cast(f2v, __builtin_elementwise_max(__builtin_bit_cast(i2v, x), __builtin_bit_cast(i2v, (f2v)(eps))));
    f2v l; i2v e;
    l.x = __builtin_amdgcn_logf(__builtin_amdgcn_frexp_mantf(xm.x)); l.y = __builtin_amdgcn_logf(__builtin_amdgcn_frexp_mantf(xm.y));
    e.x = __builtin_amdgcn_frexp_expf(xm.x); e.y = __builtin_amdgcn_frexp_expf(xm.y);
    const f2v r = (__builtin_convertvector(e, f2v) + l) * 0.693147180559945309f;
    return __builtin_elementwise_fma((f2v)(0.0f), x, r);
}
#else
__device__ __forceinline__ f4v exp4_fin_le30(f4v x)
{
    const f4v xc = __builtin_elementwise_max(x, (f4v)(-104.5f));
    const f4v t = xc * 1.44269504088896341f;
    const f4v nf = (t + 12582912.0f) - 12582912.0f;
    f4v r = vfma(nf, (f4v)(-0.693359375f), xc);
    r = vfma(nf, (f4v)(2.12194440e-4f), r);
    const f4v z = r * r;
    f4v p = (f4v)(1.9875691500e-4f);
    p = vfma(p, r, (f4v)(1.3981999507e-3f));
    p = vfma(p, r, (f4v)(8.3334519073e-3f));
    p = vfma(p, r, (f4v)(4.1665795894e-2f));
    p = vfma(p, r, (f4v)(1.6666665459e-1f));
    p = vfma(p, r, (f4v)(5.0000001201e-1f));
    p = vfma(p, z, r);
    p = p + 1.0f;
    const i4v n = __builtin_convertvector(nf, i4v);
    f4v res;
    res.x = __builtin_ldexpf(p.x, n.x); res.y = __builtin_ldexpf(p.y, n.y); res.z = __builtin_ldexpf(p.z, n.z); res.w = __builtin_ldexpf(p.w, n.w);
    // a NaN / infinite argument comes back as NaN (the clamp dropped it): (+0) * x + res as one fused operation -- for a finite x the product
    // is an exact zero, like the x - x of the scalar form in pdp_math.h, at a third of its instructions
    return vfma((f4v)(0.0f), x, res);
}

// The same four exps for arguments that are FINITE (any magnitude down to -4e6) or NaN, without the clamp and without the re-injection:
// * a NaN argument makes t, nf, r and p NaN, v_cvt_i32_f32 turns the NaN exponent into 0 and v_ldexp_f32 returns the NaN: it arrives by itself;
// * an argument below -104.5 needs no clamp: nf = rint(x log2 e) <= -151 (below -2^22 the rounding trick may leave a half-integer, the
//   conversion truncates it), r = x - nf ln 2 stays within (-1, 1), so p < 2.8 and p * 2^n < 2^-150 rounds to +0 -- what the clamped form
//   returns there (e^-104.5 = 0.59 * 2^-150).  Between -104.5 and 30 the two forms are the same instructions.
// E2 evaluates sums of clamped logs (never infinite), so it takes this form: 5 VALU instructions and 2 packed fmas fewer per slot.
__device__ __forceinline__ f4v exp4_sum(f4v xc)
{
    const f4v t = xc * 1.44269504088896341f;
    const f4v nf = (t + 12582912.0f) - 12582912.0f;
    f4v r = vfma(nf, (f4v)(-0.693359375f), xc);
    r = vfma(nf, (f4v)(2.12194440e-4f), r);
    const f4v z = r * r;
    f4v p = (f4v)(1.9875691500e-4f);
    p = vfma(p, r, (f4v)(1.3981999507e-3f));
    p = vfma(p, r, (f4v)(8.3334519073e-3f));
    p = vfma(p, r, (f4v)(4.1665795894e-2f));
    p = vfma(p, r, (f4v)(1.6666665459e-1f));
    p = vfma(p, r, (f4v)(5.0000001201e-1f));
    p = vfma(p, z, r);
    p = p + 1.0f;
    const i4v n = __builtin_convertvector(nf, i4v);
    f4v res;
    res.x = __builtin_ldexpf(p.x, n.x); res.y = __builtin_ldexpf(p.y, n.y); res.z = __builtin_ldexpf(p.z, n.z); res.w = __builtin_ldexpf(p.w, n.w);
    return res;
}

// one exp of exp4_sum as a scalar chain (the same operations)
__device__ __forceinline__ float exp1_sum(float xc)
{
    const float t = xc * 1.44269504088896341f;
    const float nf = (t + 12582912.0f) - 12582912.0f;
    float r = __builtin_fmaf(nf, -0.693359375f, xc);
    r = __builtin_fmaf(nf, 2.12194440e-4f, r);
    const float z = r * r;
    float p = 1.9875691500e-4f;
    p = __builtin_fmaf(p, r, 1.3981999507e-3f);
    p = __builtin_fmaf(p, r, 8.3334519073e-3f);
    p = __builtin_fmaf(p, r, 4.1665795894e-2f);
    p = __builtin_fmaf(p, r, 1.6666665459e-1f);
    p = __builtin_fmaf(p, r, 5.0000001201e-1f);
    p = __builtin_fmaf(p, z, r);
    p = p + 1.0f;
    return __builtin_ldexpf(p, (int)nf);
}

__device__ __forceinline__ f4v log4_fin(f4v x, float eps)
{
    // max(x, eps) on the bit patterns: for x >= +0 the integer order is the float order, a negative x (or -0) is a negative integer and gives
    // eps like the float maximum, and whatever a NaN gives is replaced by the re-injection at the end -- one instruction per element, where
    // the float maximum of a value loaded from memory costs a canonicalisation first
    const f4v xm = __builtin_bit_cast(f4v, __builtin_elementwise_max(__builtin_bit_cast(i4v, x), __builtin_bit_cast(i4v, (f4v)(eps))));
    f4v m; i4v e;
    m.x = __builtin_amdgcn_frexp_mantf(xm.x); m.y = __builtin_amdgcn_frexp_mantf(xm.y); m.z = __builtin_amdgcn_frexp_mantf(xm.z); m.w = __builtin_amdgcn_frexp_mantf(xm.w);
    e.x = __builtin_amdgcn_frexp_expf(xm.x); e.y = __builtin_amdgcn_frexp_expf(xm.y); e.z = __builtin_amdgcn_frexp_expf(xm.z); e.w = __builtin_amdgcn_frexp_expf(xm.w);
    const u4v lt = (__builtin_bit_cast(u4v, m) - 0x3f3504f3u) >> 31;
    e = e - __builtin_bit_cast(i4v, lt);
    const i4v lti = __builtin_bit_cast(i4v, lt);
    m.x = __builtin_ldexpf(m.x, lti.x); m.y = __builtin_ldexpf(m.y, lti.y); m.z = __builtin_ldexpf(m.z, lti.z); m.w = __builtin_ldexpf(m.w, lti.w);
    m = m - 1.0f;
    const f4v z = m * m;
    f4v y = (f4v)(7.0376836292e-2f);
    y = vfma(y, m, (f4v)(-1.1514610310e-1f));
    y = vfma(y, m, (f4v)(1.1676998740e-1f));
    y = vfma(y, m, (f4v)(-1.2420140846e-1f));
    y = vfma(y, m, (f4v)(1.4249322787e-1f));
    y = vfma(y, m, (f4v)(-1.6668057665e-1f));
    y = vfma(y, m, (f4v)(2.0000714765e-1f));
    y = vfma(y, m, (f4v)(-2.4999993993e-1f));
    y = vfma(y, m, (f4v)(3.3333331174e-1f));
    y = (y * m) * z;
    const f4v fe = __builtin_convertvector(e, f4v);
    y = vfma(fe, (f4v)(-2.12194440e-4f), y);
    y = vfma((f4v)(-0.5f), z, y);
    f4v r = m + y;
    r = vfma(fe, (f4v)(0.693359375f), r);
    return vfma((f4v)(0.0f), x, r);          // NaN / inf re-injection as in exp4_fin_le30 (arguments are >= +0 here: the product is +0)
}

// the same two wide (E2 takes the logs of the two values a slot's update produces): the same operation per element
__device__ __forceinline__ f2v log2_fin(f2v x, float eps)
{
    const f2v xm = __builtin_bit_cast(f2v, __builtin_elementwise_max(__builtin_bit_cast(i2v, x), __builtin_bit_cast(i2v, (f2v)(eps))));
    f2v m; i2v e;
    m.x = __builtin_amdgcn_frexp_mantf(xm.x); m.y = __builtin_amdgcn_frexp_mantf(xm.y);
    e.x = __builtin_amdgcn_frexp_expf(xm.x); e.y = __builtin_amdgcn_frexp_expf(xm.y);
    const u2v lt = (__builtin_bit_cast(u2v, m) - 0x3f3504f3u) >> 31;
    e = e - __builtin_bit_cast(i2v, lt);
    const i2v lti = __builtin_bit_cast(i2v, lt);
    m.x = __builtin_ldexpf(m.x, lti.x); m.y = __builtin_ldexpf(m.y, lti.y);
    m = m - 1.0f;
    const f2v z = m * m;
    f2v y = (f2v)(7.0376836292e-2f);
    y = __builtin_elementwise_fma(y, m, (f2v)(-1.1514610310e-1f));
    y = __builtin_elementwise_fma(y, m, (f2v)(1.1676998740e-1f));
    y = __builtin_elementwise_fma(y, m, (f2v)(-1.2420140846e-1f));
    y = __builtin_elementwise_fma(y, m, (f2v)(1.4249322787e-1f));
    y = __builtin_elementwise_fma(y, m, (f2v)(-1.6668057665e-1f));
    y = __builtin_elementwise_fma(y, m, (f2v)(2.0000714765e-1f));
    y = __builtin_elementwise_fma(y, m, (f2v)(-2.4999993993e-1f));
    y = __builtin_elementwise_fma(y, m, (f2v)(3.3333331174e-1f));
    y = (y * m) * z;
    const f2v fe = __builtin_convertvector(e, f2v);
    y = __builtin_elementwise_fma(fe, (f2v)(-2.12194440e-4f), y);
    y = __builtin_elementwise_fma((f2v)(-0.5f), z, y);
    f2v r = m + y;
    r = __builtin_elementwise_fma(fe, (f2v)(0.693359375f), r);
    return __builtin_elementwise_fma((f2v)(0.0f), x, r);
}

#endif

// cross-lane helpers without the LDS crossbar (a __shfl is a ds_bpermute round trip of ~100 cycles):
// OR of a 7-bit flag set over the wave, one ballot per bit; exchange with lane ^ 1 through a DPP quad permute.
__device__ __forceinline__ int wave_or_bits7(int bits)
{
    int r = 0;
#pragma unroll
    for (int b = 0; b < 7; ++b) r |= (__builtin_amdgcn_ballot_w64((bits >> b) & 1) != 0ull) ? (1 << b) : 0;
    return r;
}
__device__ __forceinline__ float lane_xor1(float x)
{
    return __int_as_float(__builtin_amdgcn_mov_dpp(__float_as_int(x), 0xB1 /* quad_perm:[1,0,3,2] */, 0xf, 0xf, true));
}

// select-free decoding of the packed slot words (a v_cmp + v_cndmask pair costs ~4 FMAs on gfx950)
__device__ __forceinline__ float slot_sign(uint16_t pw) { return __uint_as_float(0x3f800000u | ((uint32_t)(pw & 0x8000u) << 16)); }      // bit 15 set: -1, else +1
__device__ __forceinline__ float bit15_to_float(uint16_t w)          // bit 15 set: 1, else 0
{
    const int32_t ext = (int32_t)((uint32_t)w << 16) >> 31;
    uint32_t r;
    asm("v_and_b32 %0, 0x3f800000, %1" : "=v"(r) : "v"(ext));     // kept opaque: the optimiser would turn the mask back into a compare + select
    return __uint_as_float(r);
}
__device__ __forceinline__ float uni_f(float x) { return __int_as_float(__builtin_amdgcn_readfirstlane(__float_as_int(x))); }

// SurveyScorer tail (pdp_predict.py:174-192) with the select-free math forms: same values as d_score_from_sums (every argument
// here is finite or NaN); the three log terms of the external force take two possible values
__device__ __forceinline__ float lds_score_of(float pos, float neg, float all, float ext_sum, float Lpi, float L0)
{
    const float ef = pdp_sign(ext_sum);
    float ps = pos + ((ef == 1.0f) ? Lpi : L0);
    float ng = neg + ((ef == -1.0f) ? Lpi : L0);
    float pns = ps + ng;
    float dc = all + Lpi;
    const float bias = (2.0f * pns + dc) / 4.0f;
    ps = ps - bias; ng = ng - bias; pns = pns - bias;
    const f4v ex = exp4_fin_le30((f4v){pdp_min_c(dc - bias, 30.0f), pdp_min_c(ps, 30.0f), pdp_min_c(ng, 30.0f), pdp_min_c(pns, 30.0f)});
    dc = ex.x;
    const float q0 = ex.y - ex.w;
    const float q1 = ex.z - ex.w;
    const float total = pdp_safe_log_fin((q0 + q1) + dc, PDP_SCORER_EPS);
    const float l1 = pdp_safe_log_fin(q1, PDP_SCORER_EPS), l0 = pdp_safe_log_fin(q0, PDP_SCORER_EPS);
    return pdp_expf_fin_le30(pdp_min_c(l1 - total, 30.0f)) - pdp_expf_fin_le30(pdp_min_c(l0 - total, 30.0f));
}

// What the cold, out-of-line routines of the LDS-resident solver share with the kernel.  A non-kernel function that names a __shared__
// variable (or the dynamic LDS array) finds it through a table in memory, reads blockDim from the dispatch packet and spills what it
// touches of the caller's registers: ~7 600 cycles per call on the headline batch, as much as the routine's own work.  So the kernel
// hands over the LDS byte offsets of the instance image and of this block (laundered through an asm, or the optimiser would propagate the
// symbol back in), the workgroup size, and takes every result back in the return value -- no pointer to a local (it would live in scratch).
struct ColdShared {
    unsigned long long key;          // arg-max key of the decimation (ds_max_u64)
    int flags, found;                // OR-reductions of the decimation
    float is_sat;                    // SATProblem._is_sat of the instance (solver.py:251-252)
    int red[PDP_RED_SMALL];          // scratch of the general simplification routines
};
typedef __attribute__((address_space(3))) unsigned char lds_u8;
__device__ __forceinline__ uint32_t lds_offset_of(const void *p)
{
    uint32_t off = (uint32_t)(uintptr_t)(lds_u8 *)p;
    asm volatile("" : "+s"(off));
    return off;
}
template <class T> __device__ __forceinline__ T *lds_at(uint32_t off) { return (T *)(lds_u8 *)(uintptr_t)off; }
// the general routines of pdp_device.hpp on an LDS view: workgroup size from the view, not from the dispatch packet
__device__ __forceinline__ int team_nt(const LView &I) { return I.nt; }
template <typename T, typename Op>
__device__ __forceinline__ T team_reduce(const LView &I, T v, Op op, T identity, T *scratch) { return block_reduce(v, op, identity, scratch, I.nt); }
// (HIP's __syncthreads_or reads the three block dimensions from memory)
__device__ __forceinline__ int team_any(const LView &I, int x) { return block_reduce(x ? 1 : 0, OpOrI(), 0, I.red, I.nt); }

// Ghost sweep of an instance that leaves inactive with iterations still to come (see d_ghost_bad and k_sp_solve: would the reference's masked
// sweep over its frozen state, or the scorer on its frozen surveys, produce a non-finite value?), on the LDS image right behind the write-back:
// the arithmetic of d_ghost_bad, statement by statement, on what the exit path just stored -- q_u (parked in the q_u array by the caller), the
// final surveys, the edge-mask bit of the clause word, the force of the slot word.  X, Y, S, P, N are free there.  Cold: once per leaving
// instance, while the CU's other workgroup runs on (the separate launch it replaces ran behind the call's last chunk: ~57 us of every
// headline step for a handful of instances).  Returns 1 if something was non-finite.
__device__ __noinline__ int lds_ghost_bad(uint32_t smem_off, uint32_t cold_off, int nt, int n, int m, int ne, int fin_is_b, float pi, int has_force, int check_score, int vmask)
{
    ColdShared *const cs = lds_at<ColdShared>(cold_off);
    const LdsArrays L = carve_all(lds_at<unsigned char>(smem_off), n, m, ne);
    const int tid = threadIdx.x;
    const float *const Eta = fin_is_b ? L.EB : L.EA;
    float *const s0 = L.X, *const s3 = L.Y;
    const float L0h = pdp_safe_log(1.0f - pi * 0.0f, PDP_SP_EPS), L1h = pdp_safe_log(1.0f - pi * 1.0f, PDP_SP_EPS);
    int bad = 0;
    if (tid == 0) cs->found = 0;
    for (int c = tid; c < m; c += nt) {
        float acc = 0.0f;
        for (int k = L.f_ptr[c]; k < L.f_ptr[c + 1]; ++k) {
            const int p = L.e2p[k];
            float x = pdp_safe_log(L.QU[p], PDP_SP_EPS);
            x = x * ((L.pcc[p] & PC_EM) ? 1.0f : 0.0f);
            s0[p] = x; acc = acc + x;
        }
        L.S[c] = acc;
    }
    for (int v = tid; v < n; v += nt) {
        float P = 0.0f, N = 0.0f;
        for (int p = L.v_ptr[v]; p < L.v_ptr[v + 1]; ++p) {
            float y = pdp_safe_log(1.0f - Eta[p], PDP_SP_EPS);
            y = y * ((L.pcc[p] & PC_EM) ? 1.0f : 0.0f);
            s3[p] = y;
            const bool negative = (L.pvv[p] & 0x8000u) != 0;
            P = P + (negative ? 0.0f : 1.0f) * y;
            N = N + (negative ? 1.0f : 0.0f) * y;
        }
        L.Pv[v] = P; L.Nv[v] = N;
    }
    __syncthreads();
    for (int p = tid; p < ne; p += nt) {
        const uint16_t pw = L.pvv[p];
        const float agg = (0.0f + L.S[L.pcc[p] & 0x3fff]) - s0[p];
        const float eta_new = pdp_safe_exp(agg);
        const int v = pw & vmask;
        const SpOut o = d_sp_edge(slot_sign(pw), L.Pv[v], L.Nv[v], s3[p], has_force ? frc_of(pw) : 0.0f, L0h, L1h);
        if (!pdp_finite(eta_new) || !pdp_finite(o.qu) || !pdp_finite(o.qs) || !pdp_finite(o.dc)) bad = 1;
    }
    __syncthreads();
    for (int p = tid; p < ne; p += nt) s3[p] = pdp_safe_log(1.0f - Eta[p], PDP_SCORER_EPS) * (0.0f + L.af[L.pcc[p] & 0x3fff]);
    __syncthreads();
    for (int v = tid; v < n; v += nt) {
        float ext = 0.0f, pos = 0.0f, neg = 0.0f, all = 0.0f;
        for (int p = L.v_ptr[v]; p < L.v_ptr[v + 1]; ++p) {
            const uint16_t pw = L.pvv[p];
            const bool negative = (pw & 0x8000u) != 0;
            const float f = s3[p];
            ext = ext + (has_force ? frc_of(pw) : 0.0f);
            pos = pos + (negative ? 0.0f : 1.0f) * f;
            neg = neg + (negative ? 1.0f : 0.0f) * f;
            all = all + f;
        }
        const float sc = d_score_from_sums(pos, neg, all, ext, pi);
        if (check_score && sc != sc) bad = 1;
    }
    if (bad) cs->found = 1;
    __syncthreads();
    return cs->found;
}

// P6, cold: SurveyScorer + arg-max + set_variables (pdp_decimate.py:152-171).  Returns bit 0: a variable was fixed; bit 1: "coeff has an
// exact zero", bit 2: "NaN coefficient" (the speculation record); bits 3-4: the new value of `verified`.
#define DEC_FIXED 1
#define DEC_SPEC_SHIFT 1
#define DEC_VERIFIED_SHIFT 3
// scorer_src 1: Y already holds log(max(1 - eta, PDP_SP_EPS)) * edge mask of the CURRENT surveys (E2 left them for the next sweep).  For an
// active variable that is the scorer's own term -- the edge mask of its slots is the clause flag, and the two clamps only differ at
// 1 - eta == 0 (a survey of exactly 1: the next smaller value of 1 - eta is 2^-24, above both), where the propagator's log(1e-40) is
// replaced by the scorer's log(1e-10) on the fly; an inactive variable's coefficient is |score| * 0 either way (a NaN survey stays NaN
// under both masks).  The pass over the slots is skipped, and Y stays what the next sweep needs.  scorer_src 0 / 2: the terms are taken here,
// into Y (a sweep of the plain form: Y is free) or into the q_u array (a sweep that took the logs but may not read them here: that array
// holds the sweep's |delta eta|, which nobody needs any more).
template <bool FORCE>
__device__ __noinline__ int lds_decimate(uint32_t smem_off, uint32_t cold_off, int nt, int n, int m, int ne, int cur, int active, float pi, int verified, int scorer_src)
{
    unsigned char *const smem = lds_at<unsigned char>(smem_off);
    ColdShared *const cs = lds_at<ColdShared>(cold_off);
    const LdsArrays L = carve_all(smem, n, m, ne);
    constexpr int VMD = FORCE ? PV_VMASK_RF : 0x3fff;       // variable id of a slot word (with a force two of its bits are the force code)
    const int tid = threadIdx.x, lane = tid & 63;
    float *Enew = cur ? L.EA : L.EB;
    float *score = L.xv2, *assign = L.coeff;
    DEC_PROF_DECL
    // scorer, per slot: log(max(1 - eta, eps)) * active_clause (pdp_predict.py:168-172), four slots per trip as one 4-vector; the
    // remainder (fewer than 4 nt slots) is dealt in quarters to the first lanes, so that the other waves skip the trip
    float *const SL = (scorer_src == 2) ? L.QU : L.Y;     // the scorer's per-slot terms: Y (written here, or E2's logs read in place), or the q_u array
    const int y_has_logs = scorer_src == 1;
    if (!y_has_logs) {
        const int full = ne / (4 * nt), rem = ne - full * 4 * nt, quarter = (rem + 3) >> 2;
        for (int k = 0; k <= full; ++k) {
            const bool tail = k == full;
            if (tail && tid >= quarter) break;
            const int st = tail ? quarter : nt, p0 = k * 4 * nt + tid;
            int p[4]; bool in[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) { p[j] = p0 + j * st; in[j] = p[j] < ne; if (!in[j]) p[j] = p0; }
            const f4v lg = log4_fin((f4v){1.0f - Enew[p[0]], 1.0f - Enew[p[1]], 1.0f - Enew[p[2]], 1.0f - Enew[p[3]]}, PDP_SCORER_EPS);   // surveys are finite or NaN
            const float r[4] = {lg.x, lg.y, lg.z, lg.w};
#pragma unroll
            for (int j = 0; j < 4; ++j) if (in[j]) SL[p[j]] = r[j] * (0.0f + L.af[L.pcc[p[j]] & 0x3fff]);
        }
    }
    if (tid == 0) { cs->key = 0ull; cs->flags = 0; cs->found = 0; }
    __syncthreads();
    DEC_PROF_MARK(16);                                // scorer: edge logs
    const float Lpi = pdp_safe_log(1.0f - pi, PDP_SCORER_EPS), L0 = pdp_safe_log(1.0f - pi * 0.0f, PDP_SCORER_EPS);
    // the two clamped values, from the functions that produced / would produce the terms; without the substitution the key is a NaN, which equals nothing
    const float clamp_sp = y_has_logs ? log2_fin((f2v){0.0f, 0.0f}, PDP_SP_EPS).x : __builtin_nanf(""), clamp_sc = log4_fin((f4v)(0.0f), PDP_SCORER_EPS).x;
    int flags = 0;                                   // 1: a coefficient is exactly 0, 2: some coefficient is non-zero, 4: NaN coefficient
    unsigned long long key = 0ull;                   // util.sparse_argmax on (coeff - 0) + 1: larger value wins, first index wins ties
    for (int i = tid; i < n; i += nt) {
        const int v = L.vord[i];                     // degree-sorted: the lanes of a wave run loops of similar length
        float ext = 0.0f, pos = 0.0f, neg = 0.0f, all = 0.0f;
        auto acc = [&](float f, uint16_t pw, float frc) {
            const bool ng = (pw & 0x8000) != 0;
            f = (f == clamp_sp) ? clamp_sc : f;
            ext = ext + frc;
            pos = pos + (ng ? 0.0f : 1.0f) * f;
            neg = neg + (ng ? 1.0f : 0.0f) * f;
            all = all + f;
        };
        int p = L.v_ptr[v];
        const int bnd = L.v_ptr[v + 1];
        for (; p + 7 < bnd; p += 8) {                // all loads of a batch first: one LDS round trip per eight edges
            float f[8], fr[8]; uint16_t pw[8];
#pragma unroll
            for (int j = 0; j < 8; ++j) { f[j] = SL[p + j]; pw[j] = L.pvv[p + j]; fr[j] = FORCE ? frc_of(pw[j]) : 0.0f; }
#pragma unroll
            for (int j = 0; j < 8; ++j) acc(f[j], pw[j], fr[j]);
        }
        for (; p + 3 < bnd; p += 4) {
            float f[4], fr[4]; uint16_t pw[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) { f[j] = SL[p + j]; pw[j] = L.pvv[p + j]; fr[j] = FORCE ? frc_of(pw[j]) : 0.0f; }
#pragma unroll
            for (int j = 0; j < 4; ++j) acc(f[j], pw[j], fr[j]);
        }
        for (; p < bnd; ++p) { const uint16_t pw1 = L.pvv[p]; acc(SL[p], pw1, FORCE ? frc_of(pw1) : 0.0f); }
        const float sc = lds_score_of(pos, neg, all, ext, Lpi, L0);
        const float co = (pdp_abs(sc) * L.av[v]) * 1.0f;
        score[v] = sc; L.coeff[v] = co;
        L.flag_v[v] = 0;                             // candidate marks of the neighbourhood path below
        if (co == 0.0f) flags |= 1;
        if (co != 0.0f) flags |= 2;
        if (co != co) flags |= 4;
        // (keys need NaN-free values: with a NaN coefficient the winner is not used)
        const unsigned long long kv = argkey((co - 0.0f) + 1.0f, v);
        key = kv > key ? kv : key;
    }
    // one reduction for the flags and the arg-max: DPP inside the wave, one ds_or / ds_max_u64 per wave, one barrier (which also
    // publishes score / coeff)
    flags = wave_reduce(flags, OpOrI(), 0);
    key = wave_max_u64(key);
    if (lane == 63) { if (flags) atomicOr(&cs->flags, flags); if (key) atomicMax(&cs->key, key); }
    __syncthreads();
    flags = UNI(cs->flags);
    const int z3 = flags & 1, anynz = (flags >> 1) & 1, cn = (flags >> 2) & 1;
    int ret = ((z3 ? 1 : 0) | (cn ? 2 : 0)) << DEC_SPEC_SHIFT;
    DEC_PROF_MARK(17);                                // scorer: per-variable sums + score + flag / arg-max reduction
    const unsigned long long kwin = cs->key;
    const int li = UNI(argkey_index(kwin));
    if (!(active && anynz && !cn && li >= 0)) return ret | (verified << DEC_VERIFIED_SHIFT);
    const float sgn_li = pdp_sign(score[li]);
    DEC_PROF_MARK(18);                                // arg-max
    const LView I = make_lview(L, 0, n, m, ne, nt, cs->red, VMD);  // (the view's batch id is unused)
    SimplifyScratch ss;
    ss.assign = assign; ss.deg = reinterpret_cast<int32_t *>(L.xv1); ss.sdeg = reinterpret_cast<int32_t *>(L.xv2);
    ss.flag_v = L.flag_v; ss.flag_f = reinterpret_cast<uint8_t *>(L.S); ss.flag_f2 = ss.flag_f + ((m + 15) & ~15); ss.red = cs->red;

    // ---- fast path ------------------------------------------------------------------------------------------------
    // A problem that went through simplify() has no active unit clause and no active pure variable (both loops of
    // solver.py:228-273 / :180-203 run to a fix-point, and peeling never shrinks a clause that stays active).  Under that
    // invariant fixing ONE variable can only create unit clauses among its own clauses and pure variables among the
    // variables of the clauses it satisfied, so the full-instance sweeps of the reference shrink to the neighbourhood of
    // the variable.  The invariant is checked once per launch (the state may have been written from outside); anything
    // the neighbourhood scan finds is handed to the general routines, which redo the reference's sweeps.
    if (verified == 0) {
        int bad = 0;
        for (int c = tid; c < m; c += nt) {
            if (L.af[c] == 1.0f) {
                float deg = 0.0f;
                for (int k = L.f_ptr[c]; k < L.f_ptr[c + 1]; ++k) deg = deg + L.av[L.pvv[L.e2p[k]] & VMD];
                if (deg == 1.0f) bad = 1;
            }
        }
        for (int v = tid; v < n; v += nt) {
            if (L.av[v] == 1.0f) {
                int d = 0, sd = 0;
                int p = L.v_ptr[v];
                const int bnd = L.v_ptr[v + 1];
                for (; p + 3 < bnd; p += 4) {
                    const uint16_t c0 = L.pcc[p], c1 = L.pcc[p + 1], c2 = L.pcc[p + 2], c3 = L.pcc[p + 3];
                    const int a0 = L.af[c0 & 0x3fff] == 1.0f, a1 = L.af[c1 & 0x3fff] == 1.0f, a2 = L.af[c2 & 0x3fff] == 1.0f, a3 = L.af[c3 & 0x3fff] == 1.0f;
                    d += a0 + a1 + a2 + a3;
                    sd += ((L.pvv[p] & 0x8000) ? -a0 : a0) + ((L.pvv[p + 1] & 0x8000) ? -a1 : a1) + ((L.pvv[p + 2] & 0x8000) ? -a2 : a2) + ((L.pvv[p + 3] & 0x8000) ? -a3 : a3);
                }
                for (; p < bnd; ++p) { const int a0 = L.af[L.pcc[p] & 0x3fff] == 1.0f; d += a0; sd += (L.pvv[p] & 0x8000) ? -a0 : a0; }
                if (d == (sd < 0 ? -sd : sd)) bad = 1;
            }
        }
        verified = block_reduce(bad, OpOrI(), 0, cs->red, nt) ? 2 : 1;
        DEC_PROF_MARK(19);                            // fix-point verification (once per call)
    }
    ret |= DEC_FIXED | (verified << DEC_VERIFIED_SHIFT);
    if (verified == 1) {
        const int a = L.v_ptr[li], deg_li = L.v_ptr[li + 1] - a;
        const bool neg_li = sgn_li < 0.0f;
        // _set_variable_core for a one-hot assignment: a clause is switched off iff one of its literals of `li` is satisfied
        for (int j = tid; j < deg_li; j += nt) {
            const int p = a + j;
            const int c = L.pcc[p] & 0x3fff;
            const bool satisfied = ((L.pvv[p] & 0x8000) != 0) == neg_li;
            if (satisfied && L.af[c] == 1.0f) {
                L.af[c] = 0.0f;
                for (int k = L.f_ptr[c]; k < L.f_ptr[c + 1]; ++k) {
                    const int u = L.pvv[L.e2p[k]] & VMD;
                    if (u != li && L.av[u] == 1.0f) L.flag_v[u] = 1;      // lost a clause: may have become pure
                }
            }
        }
        if (tid == 0) { L.av[li] = 0.0f; L.sol[li] = (sgn_li + 1.0f) / 2.0f; }
        __syncthreads();
        DEC_PROF_MARK(20);                            // set the variable, switch its satisfied clauses off
        int single = 0;
        for (int j = tid; j < deg_li; j += nt) {
            const int c = L.pcc[a + j] & 0x3fff;
            if (L.af[c] == 1.0f) {
                float deg = 0.0f;
                for (int k = L.f_ptr[c]; k < L.f_ptr[c + 1]; ++k) deg = deg + L.av[L.pvv[L.e2p[k]] & VMD];
                if (deg == 1.0f) single = 1;
            }
        }
        int pure = 0;                                       // (both scans only read the state step 1 left: one reduction for the two)
        for (int v = tid; v < n; v += nt) {
            if (L.flag_v[v]) {
                int d = 0, sd = 0;
                int p = L.v_ptr[v];
                const int bnd = L.v_ptr[v + 1];
                for (; p + 3 < bnd; p += 4) {
                    const uint16_t c0 = L.pcc[p], c1 = L.pcc[p + 1], c2 = L.pcc[p + 2], c3 = L.pcc[p + 3];
                    const int a0 = L.af[c0 & 0x3fff] == 1.0f, a1 = L.af[c1 & 0x3fff] == 1.0f, a2 = L.af[c2 & 0x3fff] == 1.0f, a3 = L.af[c3 & 0x3fff] == 1.0f;
                    d += a0 + a1 + a2 + a3;
                    sd += ((L.pvv[p] & 0x8000) ? -a0 : a0) + ((L.pvv[p + 1] & 0x8000) ? -a1 : a1) + ((L.pvv[p + 2] & 0x8000) ? -a2 : a2) + ((L.pvv[p + 3] & 0x8000) ? -a3 : a3);
                }
                for (; p < bnd; ++p) { const int a0 = L.af[L.pcc[p] & 0x3fff] == 1.0f; d += a0; sd += (L.pvv[p] & 0x8000) ? -a0 : a0; }
                if (d == (sd < 0 ? -sd : sd)) pure = 1;
            }
        }
        // (`single` / `pure` are rare: the lanes that found one raise the bit themselves, one barrier joins them)
        if (single) atomicOr(&cs->found, 1);
        if (pure) atomicOr(&cs->found, 2);
        __syncthreads();
        const int found = UNI(cs->found);
        DEC_PROF_MARK(21);                            // unit-clause / pure-variable scans
#ifdef PDP_PHASE_PROF
        if (threadIdx.x == 0) { if (found & 1) atomicAdd(&g_phase_cycles[22], 1ull); else if (found & 2) atomicAdd(&g_phase_cycles[23], 1ull); }
#endif
        if (found & 1) d_simplify(I, ss, &cs->is_sat);      // a unit clause: the general routines redo the reference's sweeps
        else if (found & 2) d_peel(I, ss);
        return ret;
    }

    // ---- general path: the reference's sweeps over the whole instance -------------------------------------------------
    for (int v = tid; v < n; v += nt) assign[v] = 0.0f;
    __syncthreads();
    if (tid == 0) assign[li] = sgn_li;
    __syncthreads();
    d_set_variable_core(I, ss);
    d_simplify(I, ss, &cs->is_sat);
    return ret;
}

// Reinforce triple, cold: the force update of ReinforceDecimator.forward (pdp_decimate.py:218-232: SurveyScorer on the new surveys and
// the OLD force, force <- sign(score) on every edge of the instance) when `do_force`, then ReinforcePredictor (pdp_predict.py:221-226:
// sum of the force over the variable's edges > 0) and _update_solution (solver.py:388-399).  A variable's slots are contiguous and
// read by its own thread only, so the new force is written in the same pass; the old force goes to X (free between E2 and the next E1):
// the write-back rebuilds q_s / q_dc of the last sweep, which read the old force.  Returns 1 if a score was NaN.
// `logs`: the sweep left the NEXT sweep's logs in X / Y (E2's fused form) and |delta eta| -- dead behind P5 -- in the q_u array; they must survive
// this step.  1: the logs were taken under the edge mask: for a slot of an ACTIVE variable Y = log(max(1 - eta, 1e-40)) * mask IS the scorer's
// log(max(1 - eta, 1e-10)) * clause flag except at a survey of exactly 1 (substituted on the fly, as lds_decimate does); an inactive variable's
// terms are computed where they are used.  2: no edge mask yet (Y is not masked): the terms go to the q_u array.  Either way the force the
// sweep read is parked in the q_u array instead of X (the exit path asks qu_is_delta where it is).  0: X / Y are this step's scratch.
__device__ __noinline__ int lds_reinforce_step(uint32_t smem_off, uint32_t cold_off, int nt, int n, int m, int ne, int cur, float pi, int do_force, int logs)
{
    ColdShared *const cs = lds_at<ColdShared>(cold_off);
    const LdsArrays L = carve_all(lds_at<unsigned char>(smem_off), n, m, ne);
    const int tid = threadIdx.x;
    int bad = 0;
    const float *const Enew = cur ? L.EA : L.EB;
    float *const SL = (logs == 2) ? L.QU : L.Y;           // the scorer's per-slot terms
    if (do_force && logs != 1) {
        for (int p = tid; p < ne; p += nt)
            SL[p] = pdp_safe_log_fin(1.0f - Enew[p], PDP_SCORER_EPS) * (0.0f + L.af[L.pcc[p] & 0x3fff]);
        __syncthreads();
    }
    // the two clamped values (without the substitution the key is a NaN, which equals nothing)
    const float clamp_sp = (logs == 1) ? pdp_safe_log_fin(0.0f, PDP_SP_EPS) : __builtin_nanf(""), clamp_sc = pdp_safe_log_fin(0.0f, PDP_SCORER_EPS);
    const float Lpi = pdp_safe_log(1.0f - pi, PDP_SCORER_EPS), L0 = pdp_safe_log(1.0f - pi * 0.0f, PDP_SCORER_EPS);
    uint32_t *const codev = reinterpret_cast<uint32_t *>(L.xv2);      // per-variable force code of a renewal (xv2 is free outside the decimation)
    for (int i = tid; i < n; i += nt) {
        const int v = L.vord[i];
        const int a = L.v_ptr[v], bnd = L.v_ptr[v + 1];
        float pred;
        if (do_force) {
            // the only ORDERED part of the renewal is the score's sums (ascending slot = ascending edge id); writing the new force to the
            // variable's slots is a slot-parallel pass below, and the predictor's sum of the new force is deg * sign exactly
            float ext = 0.0f, pos = 0.0f, neg = 0.0f, all = 0.0f;
            const bool in_place = logs == 1 && L.av[v] == 1.0f;       // (an inactive variable's slots carry mask 0 in Y, the scorer wants the clause flag)
            for (int p = a; p < bnd; ++p) {
                float f = SL[p];
                if (logs == 1) f = in_place ? ((f == clamp_sp) ? clamp_sc : f) : pdp_safe_log_fin(1.0f - Enew[p], PDP_SCORER_EPS) * (0.0f + L.af[L.pcc[p] & 0x3fff]);
                const uint16_t pw = L.pvv[p];
                const bool ng = (pw & 0x8000) != 0;
                ext = ext + frc_of(pw);
                pos = pos + (ng ? 0.0f : 1.0f) * f;
                neg = neg + (ng ? 1.0f : 0.0f) * f;
                all = all + f;
            }
            const float sc = lds_score_of(pos, neg, all, ext, Lpi, L0);
            if (sc != sc) bad = 1;
            const float sg = 0.0f + pdp_sign(sc);                   // torch.sign(NaN) is 0
            codev[v] = frc_enc(sg) << PV_FRC_SHIFT;
            pred = (sg > 0.0f && bnd > a) ? 1.0f : 0.0f;
        } else {
            float ext = 0.0f;
            for (int p = a; p < bnd; ++p) ext = ext + frc_of(L.pvv[p]);
            pred = (ext > 0.0f) ? 1.0f : 0.0f;
        }
        const float av = L.av[v];
        if (av == 1.0f) L.sol[v] = av * pred + (1.0f - av) * L.sol[v];      // only active variables take the prediction (solver.py:395-397)
    }
    if (do_force) {
        __syncthreads();
        // mask * sign + (1 - mask) * old with mask == 1 (old is finite here); X -- the q_u array when X holds logs -- keeps the force the last sweep read
        float *const keep = logs ? L.QU : L.X;
        for (int p = tid; p < ne; p += nt) {
            const uint16_t pw = L.pvv[p];
            keep[p] = frc_of(pw);
            L.pvv[p] = (uint16_t)((pw & ~PV_FRC_MASK) | codev[pw & PV_VMASK_RF]);
        }
    }
    return block_reduce(bad, OpOrI(), 0, cs->red, nt);
}

// REPLAY: the poison-replay pass over ctl->replay_count listed instances (a separate instantiation, so that profilers list
// the two passes under different names)
// RF: the Reinforce triple (ReinforceDecimator + ReinforcePredictor, pdp_decimate.py:202-234) instead of the sequential decimator: no
// survey gate, no counters, no decimation; convergence (`max <= 0.01`) de-activates the instance, a shared coin per iteration decides
// whether the force is renewed.  Under the NaN poison the gate's batch-wide maximum is NaN: nobody leaves through the gate.
// LISTED (pass 1 of mixed batches): the instance comes from sp.fit_list, by ticket -- a separate instantiation, because an instance id that is
// not blockIdx.x costs a scalar register for the whole kernel and this kernel has none to spare (+3 % VALU instructions from the spill)
#ifndef PDP_SOLVE_WAVES_PER_SIMD
#define PDP_SOLVE_WAVES_PER_SIMD 4
#endif
template <bool FORCE, bool REPLAY, bool RF = false, bool LISTED = false>
__global__ void __launch_bounds__(1024, PDP_SOLVE_WAVES_PER_SIMD) k_sp_solve_lds(PView pv_, SolveParams sp)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    __shared__ float redf[PDP_RED_SMALL];
    __shared__ int redi[PDP_RED_SMALL];

    const int tid = threadIdx.x, nt = blockDim.x;
    const int lane = tid & 63, wid = tid >> 6, nw = nt >> 6;
    SolveCtl *const ctl = sp.ctl;
    if (sp.call->stop) return;                                                 // every instance went inactive in an earlier chunk
    // Which instance: normally blockIdx.x.  The hardware deals workgroups to the XCDs round-robin, so an XCD that is slower than the
    // others -- in a mixed batch one of them also hosts the team of a big instance -- makes a static launch wait for its eighth of
    // the instances.  With sp.lds_tickets every workgroup draws the next instance when it STARTS and the grid is over-provisioned:
    // the slow XCD simply gets through fewer tickets, and the workgroups left over when they run out leave at once.
    int index = (int)blockIdx.x;
    if constexpr (LISTED) {
        static_assert(!REPLAY, "the replay pass has its own list");
        if (sp.lds_tickets) {
            __shared__ int s_ticket;
            if (tid == 0) s_ticket = (int)atomicAdd(&ctl->ticket, 1u);
            __syncthreads();
            index = __builtin_amdgcn_readfirstlane(s_ticket);
            if (index >= sp.lds_tickets) return;
        }
        index = __builtin_amdgcn_readfirstlane(sp.fit_list[index]);
    }
    if (REPLAY && (!ctl->do_replay || (uint32_t)index >= ctl->replay_count)) return;
    if (!FORCE && sp.call->force_seen) return;              // the caller's force column is not all zeros: this instantiation must not run (the host reruns)
    const Inst G = load_inst(pv_, REPLAY ? __builtin_amdgcn_readfirstlane(sp.inst_list[index]) : index);
    __shared__ int s_inst;                                  // listed instance id, parked for the final writes (a scalar register less across the loop)
    if ((LISTED || REPLAY) && tid == 0) s_inst = G.b;
    const int n = G.n, m = G.m, ne = G.e;
    const LdsArrays L = carve_all(smem, n, m, ne);
    constexpr int VM = FORCE ? PV_VMASK_RF : 0x3fff;        // variable id of a slot word (with a force: two of its bits are the force code)
    float *const QU = L.QU, *const X = L.X, *const Y = L.Y;
    uint16_t *const pvv = L.pvv, *const pcc = L.pcc;
    const BlobLayout BL = blob_layout(n, m, ne);
    const char *const din = sp.dyn_in + sp.dyn_off[G.b];
    char *const dout = sp.dyn_out + sp.dyn_off[G.b];
    const DynHeader hdr = *reinterpret_cast<const DynHeader *>(din + BL.hdr);
    if (hdr.done) {
        // finished in an earlier launch (its outputs are in the caller's arrays already): carry the record forward
        if (tid == 0) {
            *reinterpret_cast<DynHeader *>(dout + BL.hdr) = hdr;
            if (hdr.perm_zero) atomicMin(&ctl->perm_zero, 0u);
            if (!REPLAY) sp.last_event[G.b] = -1;
        }
        return;
    }

    // ---- load: the instance record, global memory -> LDS without a register round trip (LDS-DMA), every piece in flight at once ------------
    // (round 6.  Eleven copy loops of the form `lds[i] = global[i]` compile to load / s_waitcnt vmcnt(0) / ds_write each: eleven DEPENDENT
    //  round trips to the records, ~31 000 workgroup-cycles per instance and launch -- what tools/phase_prof.py showed as "E1", 7.6 % of a
    //  call.  global_load_lds_dwordx4 writes LDS at [wave-uniform base + lane x 16], which is exactly a coalesced copy; it has no result
    //  register, so nothing waits until the one s_waitcnt in front of the barrier.)
    {
        const char *const stt = sp.stat + sp.stat_off[G.b];
        auto dma16 = [&](void *dst, const char *src, size_t bytes) {
            const int n16 = (int)((bytes + 15) >> 4);
            for (int i = tid; i < n16; i += nt)
                __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)(src + (size_t)i * 16),
                                                 (__attribute__((address_space(3))) void *)(reinterpret_cast<uint4 *>(dst) + (i - lane)), 16, 0, 0);
        };
        if constexpr (!FORCE) dma16(pvv, stt + BL.pvv, (size_t)ne * 2);
        dma16(L.e2p, stt + BL.e2p, (size_t)ne * 2);
        dma16(L.v_ptr, stt + BL.vptr, (size_t)(n + 1) * 2); dma16(L.f_ptr, stt + BL.fptr, (size_t)(m + 1) * 2);
        dma16(L.vord, stt + BL.vord, (size_t)n * 2);
        dma16(QU, din + BL.QU, (size_t)ne * 4); dma16(L.EA, din + BL.E, (size_t)ne * 4); dma16(pcc, din + BL.pcc, (size_t)ne * 2);
        dma16(L.af, din + BL.af, (size_t)m * 4); dma16(L.av, din + BL.av, (size_t)n * 4); dma16(L.sol, din + BL.sol, (size_t)n * 4);
        if constexpr (FORCE) {                                                 // variable word | force code of the slot
            const uint16_t *spv = reinterpret_cast<const uint16_t *>(stt + BL.pvv);
            for (int p = tid; p < ne; p += nt) pvv[p] = (uint16_t)(spv[p] | (frc_enc(sp.frc_in[G.e0 + p]) << PV_FRC_SHIFT));
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");                       // this wave's DMAs have landed; the barrier below publishes them
    }
    __shared__ __attribute__((aligned(16))) ColdShared s_cold;  // what the out-of-line routines share with the kernel (incl. SATProblem._is_sat)
    constexpr int SPEC_LOCAL = 64;
    __shared__ uint8_t s_spec_used[SPEC_LOCAL], s_spec_zero[SPEC_LOCAL];
    __shared__ int s_flag_or[2];                            // P5: the OR of the waves' flag sets, one word per sweep parity
    __shared__ int s_sat_count;                             // P8: satisfied clauses, summed over the waves
    __shared__ uint32_t s_risk;                             // smallest normalisation q_u + q_s + q_dc of the launch's last sweep (float bits, >= +0)
    __shared__ int s_poison;                                // first poisoned sweep of the chunk as this workgroup knows it (see `adopt`)
    if (tid < SPEC_LOCAL) { s_spec_used[tid] = 0; s_spec_zero[tid] = 0; }
    if (tid == 0) { s_cold.is_sat = hdr.is_sat; s_flag_or[0] = 0; s_flag_or[1] = 0; s_sat_count = 0; s_risk = 0xffffffffu; }
    __syncthreads();

    // P4's work item of this lane -- (variable, half row) -- is topology: fetched once per launch instead of through two dependent LDS round
    // trips (vord -> v_ptr) in every sweep, when every item has a lane of its own (2 n <= threads).  Two registers: lo | hi, v | degree.
    const bool p4_cached = 2 * n <= nt;
    uint32_t p4_lohi = 0u, p4_vdeg = 0u;
    if (p4_cached && tid < 2 * n) {
        const int v = L.vord[tid >> 1], h = tid & 1;
        const int a = L.v_ptr[v], bnd = L.v_ptr[v + 1];
        const int half = (bnd - a + 1) >> 1;
        p4_lohi = (uint32_t)(a + h * half) | ((uint32_t)(h ? bnd : a + half) << 16);
        p4_vdeg = (uint32_t)v | ((uint32_t)(bnd - a) << 16);
    }
    // (the same for R1 -- the slot range of a lane's variable row in one register, clause c at the by-clause positions 3 c .. 3 c + 2 when every
    //  clause has three literals -- was measured in round 5: no gain, the row sums' look-ups are not what the phase waits for)
    int active = hdr.active ? 1 : 0;
    int has_prev = sp.has_prev;
    int last_event = -1;
    int prev_from_global = (sp.has_prev && sp.prev_slots) ? 1 : 0;     // later launches: the previous surveys are the loaded ones
    int use_em = sp.has_edge_mask, last_use_em = 0, em_dirty = 0;
    float cnt = hdr.cnt;
    int iters = 0, did_prop = 0, violation = 0, cur = 0;
    int logs_ready = 0;                      // X / Y hold the logs of the current q_u / surveys already: the last sweep's E2 left them (see E2)
    int qu_is_delta = 0;                     // the q_u array holds the last sweep's |delta eta| (that sweep took the logs): q_u itself is formed at the exit
    int mask_fix = 0;                        // ... taken under the edge mask of before a refresh: the next sweep multiplies the new one in
    int nsat = (int)hdr.nsat_p1 - 1;         // clauses satisfied by `sol` (-1: not counted yet); it only changes with a decimation, so later launches inherit it
    int rf_last_flip = 0;                    // Reinforce: the force was renewed after the last sweep (X holds the one that sweep read)
    int simplified = (int)hdr.simplified;   // 0: unknown, 1: the state is a simplify() fix-point (checked at the first decimation of a call), 2: it is not
    const bool other_rows = n < pv_.V;
    const float pi = sp.pi, tol = sp.tol, t_max = sp.t_max;
    const int T = sp.T;
    const int poison_from = REPLAY ? ctl->poison_from : (sp.call->poisoned_all ? 0 : 0x7fffffff);
    // Pass 1 of a chunk runs while other workgroups of the same launch may already have met the batch's first NaN survey.  What a workgroup
    // does under the poison differs from what it does without only in its EVENTS (which `last_event` records, and the replay pass redoes
    // from the poison on), so a workgroup may take any sweep >= the launch's final first-NaN sweep as poisoned at once: thread 0 looks at
    // the chunk's record when the workgroup starts (a launch is ~10 rounds of workgroups), and from the first sweep >= the value it saw the
    // instance runs poisoned -- it then has no event at or behind the poison and needs no replay.  The value seen can only be too large (the
    // minimum is still being formed): then those sweeps run unpoisoned as before and the replay covers them.  Which workgroups see what depends
    // on the dispatch order; the results do not.  (Looking once per sweep replays another 10 % fewer instances and costs 0.8 us per sweep --
    // an agent-scope load is a round trip past the L2, and the next barrier waits for it: +0.43 ms per call against -0.24 ms of replay.)
    const bool adopt = !REPLAY && poison_from == 0x7fffffff && !sp.isolate && sp.adopt_poison;
    if (tid == 0) {
        int from = poison_from;
        if (adopt) { const uint32_t seen = __hip_atomic_load(&ctl->nan_iter, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); if (seen < (uint32_t)from) from = (int)seen; }
        s_poison = from;
    }
    __syncthreads();
    // log(max(1 - pi * [force == +-s], eps)): two possible values per kernel (pdp_propagate.py:197,201)
    const float L0 = pdp_safe_log(1.0f - pi * 0.0f, PDP_SP_EPS), L1 = pdp_safe_log(1.0f - pi * 1.0f, PDP_SP_EPS);

    PROF_DECL
    PROF_MARK(0);                                            // load
    for (int t = 0; t < T; ++t) {
        if (!active) break;
        iters = t + 1;
        float *const Eold = cur ? L.EB : L.EA, *const Enew = cur ? L.EA : L.EB;
        // ---- E1: per-slot logs, two slots per trip (independent chains for the scheduler / packed fp32 ops) -------------
        // Full trips take slots (p, p + nt); the remainder (fewer than 2 nt slots) is split in two halves that the first `half` lanes take
        // the same way, so that the waves beyond it skip the trip: 2 520 slots on 512 threads are two full trips and 472 slots that four
        // waves take as pairs, where eight waves took one slot each at the price of a pair.  (Four slots per trip -- every wave one
        // trip, two waves the remainder -- was measured in round 5: +3 % on the launch; the phase ends with its slowest wave, and that
        // form gives two waves four 4-vectors where this one gives four waves three.)
        // A sweep that follows a plain sweep of this launch finds its logs in place (E2 took them of the values it produced) and has no E1.
        if (!logs_ready) {
        if (!PROF_SKIP(1)) {
            const int full = ne / (2 * nt), rem = ne - full * 2 * nt, half = (rem + 1) >> 1;
            for (int k = 0; k <= full; ++k) {
                const bool tail = k == full;
                if (tail && tid >= half) break;
                const int p0 = k * 2 * nt + tid, p1 = p0 + (tail ? half : nt);
                const bool has1 = p1 < ne;
                const int q1 = has1 ? p1 : p0;
                // (as four scalar chains, like E2's: +0.4 % on the launch -- two slots' logs have the packed form's parallelism anyway)
                const f4v lg = log4_fin((f4v){QU[p0], QU[q1], 1.0f - Eold[p0], 1.0f - Eold[q1]}, PDP_SP_EPS);
                float x0 = lg.x, x1 = lg.y, y0 = lg.z, y1 = lg.w;
                if (use_em) {
                    uint16_t c0 = pcc[p0], c1 = pcc[q1];
                    const float em0 = bit15_to_float(c0), em1 = bit15_to_float(c1);
                    x0 = x0 * em0; y0 = y0 * em0; x1 = x1 * em1; y1 = y1 * em1;
                    if (em_dirty) {
                        pcc[p0] = (uint16_t)((c0 & ~PC_EM_USED) | ((c0 & PC_EM) ? PC_EM_USED : 0));
                        if (has1) pcc[p1] = (uint16_t)((c1 & ~PC_EM_USED) | ((c1 & PC_EM) ? PC_EM_USED : 0));
                    }
                }
                X[p0] = x0; Y[p0] = y0;
                if (has1) { X[p1] = x1; Y[p1] = y1; }
            }
        }
        __syncthreads();
        } else if (mask_fix) {
            // The logs are in place but a refresh changed the edge mask (a decimation clears flags, it never sets one): log * new mask is the old
            // product times the new mask -- exact, x * 1 = x and (+-0) * 0 keeps its sign --, and only the slots that lost their bit change.  The
            // bit "the mask the last propagate used" follows here, as it does in E1 (the sweep that is about to run is that propagate).
            for (int p = tid; p < ne; p += nt) {
                const uint16_t c = pcc[p];
                if (!(c & PC_EM)) { X[p] = X[p] * 0.0f; Y[p] = Y[p] * 0.0f; }
                if (((c >> 1) ^ c) & PC_EM_USED) pcc[p] = (uint16_t)((c & ~PC_EM_USED) | ((c & PC_EM) ? PC_EM_USED : 0));
            }
            __syncthreads();
            PROF_MARK(19);                                   // the mask pass behind a refresh
        }
        last_use_em = use_em; em_dirty = 0; mask_fix = 0;
        PROF_MARK(1);                                        // E1 (where it runs) + the top of the loop
        // ---- R1: per-clause sums (through e2p) and per-variable sums (contiguous), ascending edge id ------------
        {
            const uint16_t *const e2p = L.e2p, *const f_ptr = L.f_ptr, *const v_ptr = L.v_ptr;
            float *const S = L.S, *const Pv = L.Pv, *const Nv = L.Nv;
            // Work is handed out per wave in items of 64 rows.  A variable row is a sequential sum over ~|E|/n terms and costs
            // about two clause rows, so the waves that take a variable item skip the first two rounds of clause items.
            const int nvi = (n + 63) >> 6, nci = (m + 63) >> 6;
            // P / N: ordered sums of y over the positive / negative edges.  The reference adds (mask * y) for every edge; a masked-out
            // term is +-0 and never changes the running sum (which starts at +0 and therefore is never -0), so the terms of the
            // other sign are cleared with a bit mask instead of multiplied; a NaN y must reach both sums and is re-injected at the end.
            const uint16_t *const vord = L.vord;
            auto var_rows = [&](int item) {
                const int i = (item << 6) + lane;
                if (i >= n) return;
                const int v = vord[i];
                float P = 0.0f, N = 0.0f;
                const int a = v_ptr[v], bnd = v_ptr[v + 1];
                auto acc = [&](float y, uint16_t sg) {
                    const uint32_t neg = (uint32_t)((int32_t)((uint32_t)sg << 16) >> 31);       // all ones for a negative literal
                    P = P + __uint_as_float(__float_as_uint(y) & ~neg);
                    N = N + __uint_as_float(__float_as_uint(y) & neg);
                };
                int p = a;
                for (; p + 15 < bnd; p += 16) {                 // as many loads in flight per LDS round trip as the degree allows, then the ordered adds
                    float y[16]; uint16_t sg[16];
#pragma unroll
                    for (int j = 0; j < 16; ++j) { y[j] = Y[p + j]; sg[j] = pvv[p + j]; }
#pragma unroll
                    for (int j = 0; j < 16; ++j) acc(y[j], sg[j]);
                }
                for (; p + 7 < bnd; p += 8) {
                    float y[8]; uint16_t sg[8];
#pragma unroll
                    for (int j = 0; j < 8; ++j) { y[j] = Y[p + j]; sg[j] = pvv[p + j]; }
#pragma unroll
                    for (int j = 0; j < 8; ++j) acc(y[j], sg[j]);
                }
                for (; p + 3 < bnd; p += 4) {
                    float y[4]; uint16_t sg[4];
#pragma unroll
                    for (int j = 0; j < 4; ++j) { y[j] = Y[p + j]; sg[j] = pvv[p + j]; }
#pragma unroll
                    for (int j = 0; j < 4; ++j) acc(y[j], sg[j]);
                }
                {
                    float y[3]; uint16_t sg[3];
#pragma unroll
                    for (int j = 0; j < 3; ++j) { const int q = (p + j < bnd) ? p + j : a; y[j] = Y[q]; sg[j] = pvv[q]; }
#pragma unroll
                    for (int j = 0; j < 3; ++j) if (p + j < bnd) acc(y[j], sg[j]);
                }
                Pv[v] = P + 0.0f * N; Nv[v] = N + 0.0f * P;
            };
            // Clause rows are a three-level dependent LDS chain (f_ptr -> e2p -> X): a lane takes up to four rows at once so that the
            // chain is paid once per group, not once per row; the straight-line form needs 3-literal clauses in all of the lane's rows.
            auto clause_row = [&](int r) {
                float acc = 0.0f;
                const int a = f_ptr[r], bnd = f_ptr[r + 1];
                for (int k = a; k < bnd; ++k) acc = acc + X[e2p[k]];
                S[r] = acc;
            };
            // (branch-free up to the one wave-uniform test: a lane without a row in the group reads row 0 and stores to the spare
            //  slot S[m] -- the per-row exec-mask regions of the first form were most of the instructions of a group, and the clause
            //  waves are the critical path of this phase)
            const bool can_group = m > 0 && ne >= 3;
            auto clause_group = [&](int j0, int cnt) {
                int r[4], fa[4], fb[4]; bool in[4]; bool all3 = can_group;
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    r[i] = ((j0 + i) << 6) + lane; in[i] = i < cnt && r[i] < m;
                    const int rr = in[i] ? r[i] : 0;
                    fa[i] = f_ptr[rr]; fb[i] = f_ptr[rr + 1];
                }
#pragma unroll
                for (int i = 0; i < 4; ++i) all3 = all3 && (!in[i] || fb[i] - fa[i] == 3);
                if (__builtin_amdgcn_ballot_w64(!all3) == 0) {
                    uint16_t e[4][3]; float x[4][3];
#pragma unroll
                    for (int i = 0; i < 4; ++i) { e[i][0] = e2p[fa[i]]; e[i][1] = e2p[fa[i] + 1]; e[i][2] = e2p[fa[i] + 2]; }
#pragma unroll
                    for (int i = 0; i < 4; ++i) { x[i][0] = X[e[i][0]]; x[i][1] = X[e[i][1]]; x[i][2] = X[e[i][2]]; }
#pragma unroll
                    for (int i = 0; i < 4; ++i) { float acc = 0.0f; acc = acc + x[i][0]; acc = acc + x[i][1]; acc = acc + x[i][2]; S[in[i] ? r[i] : m] = acc; }
                } else {
#pragma unroll
                    for (int i = 0; i < 4; ++i) if (in[i]) clause_row(r[i]);
                }
            };
            if (!PROF_SKIP(2)) {
                if (!PROF_SKIP(32)) for (int i = wid; i < nvi; i += nw) var_rows(i);
                // clause items go to the waves without a variable item (a variable row is several times longer), in contiguous chunks
                // (letting the wave with the last, nearly empty variable item take clause items too -- 14 items over five waves instead of
                //  four -- was measured in round 5: +0.9 % on the launch)
                const int first = (nvi < nw) ? nvi : nw;
                const int helpers = nw - first;
                int j_begin, j_end, j_step;
                if (helpers > 0) { const int chunk = (nci + helpers - 1) / helpers; j_begin = (wid - first) * chunk; j_end = j_begin + chunk; j_step = 0; if (wid < first) j_end = j_begin = 0; }
                else { j_begin = wid * 4; j_end = nci; j_step = nw * 4; }
                if (j_end > nci) j_end = nci;
                if (!PROF_SKIP(64)) {
                    if (j_step == 0) { for (int j = j_begin; j < j_end; j += 4) clause_group(j, (j_end - j) < 4 ? (j_end - j) : 4); }
                    else { for (int j = j_begin; j < j_end; j += j_step) clause_group(j, (nci - j) < 4 ? (nci - j) : 4); }
                }
            }
        }
        __syncthreads();
        // (behind this barrier every wave has left the previous sweep -- a sweep without E1 has no barrier in front of R1)
        if (tid == 0) { s_flag_or[(t + 1) & 1] = 0; s_sat_count = 0; }
        PROF_MARK(2);                                        // R1
        // ---- E2: new survey, new q_u, smooth-max weights -----------------------------------------------------------
        // The next sweep's E1 would take the logs of exactly the two values a slot's update stores (q_u, the survey), in the same lane order:
        // E2 takes them here -- X / Y of the slot are this lane's own to overwrite once it has read them --, and the next sweep starts at
        // R1: no pass over the slots to fetch and unpack them again, one workgroup barrier less.  Y then no longer carries |delta eta| to P4 --
        // the q_u array does: nothing reads q_u itself while the logs are in place (the sticky NaN of `mask * new + (1 - mask) * old` comes
        // from X = log(old q_u) * mask, which is NaN exactly when the old q_u is -- log(max(x, eps)) is finite for every finite x -- and the
        // exit path forms q_u from the same sums as its two other columns).  Not in the last sweep of a launch (the record that goes to the
        // next launch holds q_u), not in a call's first sweep, and not in a Reinforce sweep that renews the force (that step uses X / Y as scratch,
        // so the sweep behind it needs q_u for its logs).  A sweep of the plain form stores q_u and hands |delta eta| over in Y as rounds 1-4 did.
        // The two forms of the loop are separate loops: nothing to branch on per trip.
        // (Reinforce: in the sweeps whose coin renews no force -- the coins of a call are drawn up front -- nothing uses X / Y as scratch either)
        const bool rf_step_now = RF && ((sp.coins[sp.chunk_start + t] < sp.dprob) || nsat < 0);
        const bool fuse_logs = (!rf_step_now || !sp.rf_no_fused_step) && (t + 1 < T) && has_prev && !prev_from_global;
        float nan_acc = 0.0f;
        const uint32_t log_em_or = use_em ? 0u : PC_EM;      // without an edge mask every slot counts
        // one slot's update: reads the slot's words, the three row sums and its own logs, stores the new survey and returns the new values
        struct SlotNew { float qu, eta, eta_old, total; uint16_t cw; };
        const float *const S_ = L.S, *const Pv_ = L.Pv, *const Nv_ = L.Nv;
        auto slot_update = [&](int p) __attribute__((always_inline)) {
            const uint16_t pw = pvv[p], cw = pcc[p];
            const int v = pw & VM, c = cw & 0x3fff;
            const float s = slot_sign(pw);
            const float eta_old = Eold[p];
            const float xp = X[p];
            const float agg = S_[c] - xp;                       // the reference's 0 + S is a no-op: a sum that starts at +0 is never -0
            const float force = FORCE ? frc_of(pw) : 0.0f;
            const float pos = Pv_[v], neg = Nv_[v];
            // The reference's (0.5 (1 + s)) * pos + (0.5 (1 - s)) * neg has coefficients 1 and 0: one product is the sum itself, the other an
            // exact zero.  R1 stores Pv / Nv so that they are never -0 (sums that start at +0), never infinite (sums of clamped logs) and NaN
            // only together (each gets 0 * the other): x + (+-0) == x then, so the expression is the SELECTED sum -- two selects on the slot's
            // sign bit instead of the coefficient arithmetic, two packed multiplies and the adds.
            const bool neg_lit = (pw & 0x8000u) != 0;
            float same = neg_lit ? neg : pos;
            same = same - Y[p];
            // without an external force both log terms are log(1) = +0: adding it can only turn a -0 into +0, which exp ignores
            if constexpr (FORCE) same = same + ((force == s) ? L1 : L0);
            float opp = neg_lit ? pos : neg;
            if constexpr (FORCE) opp = opp + ((force == -s) ? L1 : L0);
#ifndef PDP_E2_PACKED_EXPS
            // (four scalar chains: a packed fp32 instruction holds the SIMD as long as two plain ones and waits a state behind the packed step it
            //  depends on, so the 4-vector form buys nothing here -- measured in round 5: -2.3 % on the launch, the logs likewise -1.8 %)
            const float so = same + opp;
            const f4v ex = {exp1_sum(agg), exp1_sum(so), exp1_sum(same), exp1_sum(opp)};
#else
            const f4v ex = exp4_sum((f4v){agg, same + opp, same, opp});
#endif
            // mask * new + (1 - mask) * old with mask == 1: (+0) * old + new as ONE fused operation -- the product is an exact zero (or NaN),
            // so fusing rounds nothing differently; the unfused form is a multiply and an add per slot
            const float eta_new = __builtin_fmaf(1.0f - 1.0f, eta_old, ex.x);
            const float dc = ex.y;
            const float A = ex.z, Bv = ex.w;
            const float qu = A * (1.0f - Bv), qs = Bv * (1.0f - A);
            const float total = (qu + qs) + dc;
            const float qu_new = __builtin_fmaf(1.0f - 1.0f, xp, qu / total);      // (+0) * log(old q_u): NaN iff the old q_u is NaN or infinite
            Enew[p] = eta_new;
            return SlotNew{qu_new, eta_new, eta_old, total, cw};
        };
        auto slot_logs = [&](int p, const SlotNew &r) __attribute__((always_inline)) {
#ifndef PDP_E2_PACKED_LOGS
            // (the two logs as scalar chains: the survey's log runs beside the division it does not depend on, and neither waits a state between
            //  dependent packed steps -- measured in round 5 against the 2-vector form: -1.8 % on the launch)
            const float em = bit15_to_float((uint16_t)(r.cw | log_em_or));
            Y[p] = pdp_safe_log_fin(1.0f - r.eta, PDP_SP_EPS) * em;      // (needs the survey only: runs beside the division)
            X[p] = pdp_safe_log_fin(r.qu, PDP_SP_EPS) * em;
#else
            const f2v lg = log2_fin((f2v){r.qu, 1.0f - r.eta}, PDP_SP_EPS) * bit15_to_float((uint16_t)(r.cw | log_em_or));
            X[p] = lg.x; Y[p] = lg.y;
#endif
        };
        if (PROF_SKIP(4)) { }
        else if (fuse_logs) {
            // (the logs of a trip riding in the NEXT trip, so that their dependent chain of ~20 packed steps interleaves with that trip's loads, exps
            //  and division in one basic block, was measured in round 5: +3 % on the launch)
            for (int p = tid; p < ne; p += nt) {
                const SlotNew r = slot_update(p);
                QU[p] = pdp_abs(r.eta_old - r.eta) * bit15_to_float((uint16_t)(r.cw | log_em_or));   // |delta eta| * edge mask (pdp_decimate.py:135-141)
                slot_logs(p, r);
            }
        } else {
            // (a launch's last sweep is of this form: the smallest normalisation of the instance goes to the dispatch order of the next launch)
            float tmin = PDP_INF;
            for (int p = tid; p < ne; p += nt) {
                const SlotNew r = slot_update(p);
                tmin = (r.total < tmin) ? r.total : tmin;
                QU[p] = r.qu;
                if (!has_prev) {
                    // a NaN survey: with a previous survey every slot's |difference| is NaN too and P4's sums flag it (its S1 covers every slot
                    // of the instance); only the first sweep of a solve without one has to look here
                    nan_acc = __builtin_fmaf(0.0f, r.eta, nan_acc);    // stays 0 unless a survey is NaN (surveys are <= 1, never infinite)
                } else {
                    const float pe = prev_from_global ? sp.prev_slots[G.e0 + p] : r.eta_old;
                    float d = pdp_abs(pe - r.eta);
                    if (use_em) d = d * bit15_to_float(r.cw);
                    Y[p] = d;                               // the smooth-max weights exp(30 d) are only built when P4 cannot decide without them
                }
            }
            if (sp.risk && t + 1 == T) {
                tmin = wave_reduce(tmin, OpMinLess(), PDP_INF);
                if (lane == 63) atomicMin(&s_risk, __float_as_uint(tmin));      // totals are >= +0 (or NaN, which sorts behind everything)
            }
        }
        int nan_seen = (nan_acc != nan_acc) ? 1 : 0;
        did_prop = 1;
        logs_ready = fuse_logs ? 1 : 0;
        qu_is_delta = logs_ready;
        float *const Dsrc = fuse_logs ? QU : Y;              // where this sweep's |delta eta| is (dead behind P5b: the rare exact passes' scratch)
        __syncthreads();
        PROF_MARK(3);                                        // E2
        // ---- P4: per-variable smooth maxima, decided lazily.  The two batch-visible facts per instance are booleans:
        // site 0 (survey gate `max <= 1e-10`, pdp_decimate.py:127-133) and site 1 (`max |delta eta| < tolerance`, :135-146).
        // Site 0: smooth_max_v >= eta_max_v / deg_v (the largest survey carries the largest weight), so one active variable
        // with eta_max_v >= 2^-21 * deg_v proves (x + 1) - 1 >= 2^-23 > 1e-10; an exact zero of the operand exists iff some
        // variable is inactive or has only zero surveys.
        // Site 1: with d in [0, D], W = exp(30 D), S1 = sum d, S2 = sum d^2 the chord / tangent of exp give
        //   (D W + (S1 - D) + 30 (S2 - D^2)) / (deg + S1 (W - 1) / D)  <=  smooth_max_v  <=  min(D, (S1 + S2 (W - 1) / D) / (deg + 30 S1)),
        // at most one exp per VARIABLE instead of one per edge (none when D itself is below the tolerance).  A variable whose bounds straddle the tolerance (or whose D is so small
        // that the quotient may round to zero) is marked and evaluated exactly in P5b, in the reference's summation order.
        int bits = nan_seen ? 4 : 0;       // 1: site-0 operand has an exact 0, 2: site-1 operand has one, 4: NaN, 8: gate certified open,
                                           // 16: some variable proves "not converged", 32: undecided variables, 64: tiny-D variables
        {
            // two lanes per variable, each scans half of the variable's slots; every statistic here is order-free (maxima, and
            // sums that only feed certificates with a safety margin), so the halves combine through one lane exchange
            const uint16_t *const v_ptr = L.v_ptr;
            const float *const av = L.av;
            float *const amb = L.xv1;
            const float tol_lo = tol - (2e-6f + 1e-4f * pdp_abs(tol)), tol_hi = tol + (2e-6f + 1e-4f * pdp_abs(tol));
            if (!PROF_SKIP(8))
            for (int r = tid; r < 2 * n; r += nt) {
                int v, lo, hi, degi;
                if (p4_cached) { lo = (int)(p4_lohi & 0xffffu); hi = (int)(p4_lohi >> 16); v = (int)(p4_vdeg & 0xffffu); degi = (int)(p4_vdeg >> 16); }
                else {
                    v = L.vord[r >> 1];
                    const int h = r & 1, a = v_ptr[v], bnd = v_ptr[v + 1];
                    const int half = (bnd - a + 1) >> 1;
                    lo = a + h * half; hi = h ? bnd : a + half; degi = bnd - a;
                }
                float emax = 0.0f, S1 = 0.0f, S2 = 0.0f, D = 0.0f;
                if (has_prev) {
                    int p = lo;
                    // (four slots per trip -- their loads in flight together -- was measured in round 5: +3 % on the launch)
                    for (; p + 1 < hi; p += 2) {
                        const float e0 = Enew[p], e1 = Enew[p + 1], d0 = Dsrc[p], d1 = Dsrc[p + 1];
                        emax = fmaxf(emax, fmaxf(e0, e1));           // (a NaN survey is dropped here and caught by S1 below)
                        S1 += d0 + d1; S2 = fmaf(d0, d0, fmaf(d1, d1, S2)); D = fmaxf(D, fmaxf(d0, d1));
                    }
                    if (p < hi) { const float e0 = Enew[p], d0 = Dsrc[p]; emax = fmaxf(emax, e0); S1 += d0; S2 = fmaf(d0, d0, S2); D = fmaxf(D, d0); }
                } else {
                    for (int p = lo; p < hi; ++p) emax = fmaxf(emax, Enew[p]);
                }
                emax = fmaxf(emax, lane_xor1(emax));
                S1 += lane_xor1(S1); S2 += lane_xor1(S2); D = fmaxf(D, lane_xor1(D));
                const float a_v = av[v];
                const float deg = (float)degi;
                if (a_v == 0.0f || !(emax > 0.0f)) bits |= 1;
                if (a_v == 1.0f && emax >= 4.76837158203125e-7f * deg) bits |= 8;
                if (!has_prev) continue;
                float code = 0.0f;
                bool bounds = false;
                if (S1 != S1) bits |= 4;                    // a NaN survey difference (or a NaN survey: E2 leaves that to this test)
                else if (a_v == 0.0f || D == 0.0f) bits |= (0.0f < tol) ? 2 : (2 | 16);   // smooth max * active == 0 exactly
                else if (!(D >= 1e-30f)) { bits |= 64; code = 2.0f; }                      // the quotient may round to zero: exact
                else if (D * 1.0001f < tol_lo) { }                                        // smooth max <= D: below the tolerance
                else if (S1 * 0.9999f >= tol_hi * deg) bits |= 16;                        // smooth max >= mean (the weights exp(30 d) grow with d): above it, no exp
                else if (D >= 1e-15f) bounds = true;
                else { bits |= 32; code = 1.0f; }
                // One variable that proves "not converged" settles the instance (P5b and the decision only look at the undecided ones when
                // nobody did), so a wave that holds such a variable skips the bounds of its other variables -- the exp of the bounds is most of
                // this phase's instructions, and far from convergence nearly every wave holds one.
                if (__builtin_amdgcn_ballot_w64(bounds) != 0 && __builtin_amdgcn_ballot_w64((bits & 16) != 0) == 0) {
                    if (bounds) {
                        // the bounds above, cross-multiplied by D so that no division is needed
                        const float W = pdp_expf_fin_le30(30.0f * D), Wm1 = W - 1.0f;
                        const float ubn = fmaf(S2, Wm1, S1 * D), ubd = D * fmaf(30.0f, S1, deg);
                        const float lbn = D * (fmaf(D, W, S1 - D) + 30.0f * (S2 - D * D)), lbd = fmaf(S1, Wm1, D * deg);
                        if (lbn * 0.9999f >= tol_hi * lbd) bits |= 16;
                        else if (!(ubn * 1.0001f < tol_lo * ubd)) { bits |= 32; code = 1.0f; }
                    }
                }
                amb[v] = code;
            }
        }
        PROF_MARK(4);                                        // P4
        // ---- P5: one fused workgroup reduction of the flag bits ------------------------------------------------------------
        // (one LDS word per sweep parity collects the waves' sets with ds_or: one barrier and one broadcast read, where a slot per wave
        //  took two barriers and nw reads; the other parity's word is cleared by thread 0 behind this sweep's R1 barrier, after every
        //  wave has used it -- in the previous sweep -- and before the next sweep's P5)
        bits = wave_or_bits7(bits);
        if (lane == 0 && bits) atomicOr(&s_flag_or[t & 1], bits);
        // s_poison is read BEFORE the barrier: its only writer inside the loop (thread 0 in the event look below) runs between this barrier and
        // the one that closes the event look, so a read behind the barrier could see this sweep's store in one wave and not in another --
        // `poisoned` must be workgroup-uniform (it guards barriers)
        const int poison_known = UNI(s_poison);
        __syncthreads();
        bits = UNI(s_flag_or[t & 1]);                        // workgroup-uniform: keep the control flow scalar
        bool poisoned = t >= poison_known;
        // ---- P5b (rare): exact smooth max of the marked variables (util.py:282-286 + :267-275 with the global min at 0)
        if ((bits & 64) || ((bits & 32) && !(bits & 16))) {
            PROF_COUNT(9);
            int b2 = 0;
            const float *const amb = L.xv1;
            const bool need_decision = !(bits & 16);
            for (int v = tid; v < n; v += nt) {
                const float code = amb[v];
                if (code == 2.0f || (code == 1.0f && need_decision)) {
                    float num = 0.0f, den = 0.0f;
                    for (int p = L.v_ptr[v]; p < L.v_ptr[v + 1]; ++p) {
                        const float d = Dsrc[p];
                        const float c0 = pdp_expf_fin_le30(30.0f * d);
                        num = num + d * c0; den = den + c0;
                    }
                    const float rr = (num / pdp_max_c(den, 1.0f)) * L.av[v];
                    if (rr == 0.0f) b2 |= 2;
                    const float mv = (((rr - 0.0f) + 1.0f) + 0.0f) - 1.0f;
                    if (RF ? !(mv <= tol) : !(mv < tol)) b2 |= 16;
                }
            }
            b2 = wave_or_bits7(b2);
            if (lane == 0) redi[wid] = b2;
            __syncthreads();
            b2 = 0;
            for (int i = 0; i < nw; ++i) b2 |= redi[i];
            __syncthreads();
            bits |= UNI(b2);
        }
        PROF_MARK(5);                                        // P5
        float g;
        if (RF || (bits & 8)) {
            g = 1.0f;                                       // certified: the gate stays open, its exact value is never used (Reinforce has no survey gate)
        } else {
            // exact path (rare: every active variable has only vanishing surveys): util.py:282-286 + :267-275
            for (int p = tid; p < ne; p += nt) Dsrc[p] = pdp_expf_fin_le30(30.0f * Enew[p]);
            __syncthreads();
            float mm = -PDP_INF;
            for (int v = tid; v < n; v += nt) {
                float num = 0.0f, den = 0.0f;
                for (int p = L.v_ptr[v]; p < L.v_ptr[v + 1]; ++p) { const float c0 = Dsrc[p]; num = num + Enew[p] * c0; den = den + c0; }
                const float rr = (num / pdp_max_c(den, 1.0f)) * L.av[v];
                mm = pdp_max(mm, (rr - 0.0f) + 1.0f);
            }
            mm = block_reduce(mm, OpMaxNan(), -PDP_INF, redf);
            mm = uni_f(mm);
            if (other_rows) mm = pdp_max(mm, 0.0f);
            g = (mm + 0.0f) - 1.0f;
        }
        // site 1: (max_v (rr_v + 1)) - 1 < tol  <=>  every variable passes the same test on its own (both maps are monotone);
        // a NaN anywhere makes the reference's maximum NaN and the comparison false
        const bool below_tol = (n > 0) ? (!(bits & 16) && !(bits & 4)) : (RF ? ((other_rows ? -1.0f : -PDP_INF) <= tol) : ((other_rows ? -1.0f : -PDP_INF) < tol));
        const int z1 = bits & 1, z2 = (bits >> 1) & 1;
        nan_seen = (bits >> 2) & 1;
        if (nan_seen && !poisoned) {
            if (tid == 0) atomicMin(&ctl->nan_iter, (uint32_t)t);
            if (poison_from != 0x7fffffff) violation = 1;
        }
        // An event is about to happen (the only thing the poison would change): look once more whether the batch is poisoned by now -- a workgroup
        // of the first round started together with the one that meets the NaN, and its events come later in the chunk.  Cold: one sweep in fifteen.
        if (adopt && !poisoned && !sp.no_event_look && ((has_prev && (below_tol || cnt >= t_max)) || (!RF && g <= 1e-10f))) {
            if (tid == 0) { const uint32_t seen = __hip_atomic_load(&ctl->nan_iter, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); if (seen < (uint32_t)s_poison) s_poison = (int)seen; }
            __syncthreads();
            poisoned = t >= UNI(s_poison);
        }
        int conv = 0;
        int rf_changed = 0;
        if constexpr (RF) {
            // active_mask[sum_diff <= 0.01] = 0 (pdp_decimate.py:205-215); under the poison the batch-wide maximum is NaN and nobody leaves
            if (!poisoned && has_prev && below_tol) { active = 0; last_event = t; }
        } else
        if (!poisoned) {
            // an "event" is anything a NaN-poisoned batch would NOT do: pass 1 remembers the last one so that only the
            // instances with an event at or after the poison iteration have to be replayed
            if (g <= 1e-10f) { active = 0; last_event = t; }
            if (has_prev) {
                if (below_tol) { cnt = 0.0f; last_event = t; }
                conv = below_tol ? 1 : 0;
                if (cnt >= t_max) { conv = 1; cnt = 0.0f; }
                if (conv) last_event = t;
            }
        } else if (has_prev) {
            if (cnt >= t_max) { conv = 1; cnt = 0.0f; }
        }
        uint32_t used = (RF ? 0u : 1u) | (has_prev ? 2u : 0u);
        uint32_t zero = ((!RF && z1) ? 1u : 0u) | ((has_prev && z2) ? 2u : 0u);
        if (!RF && has_prev && !conv && n > 0) zero |= 4u;
        if constexpr (RF) {
            // the shared coin of this iteration (pdp_decimate.py:218); instances that are still active after the gate renew their force.
            // Predictor + _update_solution run in every iteration of the reference; their result only changes with the force.
            const int flip = (sp.coins[sp.chunk_start + t] < sp.dprob) && active;
            rf_last_flip = flip;
            if (flip || nsat < 0) {
                // (a sweep that took the next one's logs keeps them: the step reads the scorer's terms from Y in place, or builds them in the q_u array)
                const int keep_logs = fuse_logs ? (use_em ? 1 : 2) : 0;
                if (UNI(lds_reinforce_step(lds_offset_of(smem), lds_offset_of(&s_cold), nt, n, m, ne, cur, pi, flip, keep_logs)) && !nan_seen) violation = 1;    // a NaN score without a NaN survey: not expected
                rf_changed = 1;
                if (!keep_logs) logs_ready = 0;             // (X / Y were its scratch)
            }
        }
        PROF_MARK(13);                                       // gate + bookkeeping of every sweep
        // ---- P6: decimation (rare, out of line) ------------------------------------------------------------------------------
        int decimated = 0;
        if (!RF && has_prev && conv && !poisoned && !nan_seen && !PROF_SKIP(16)) {
            // (with the next sweep's logs in Y the scorer reads them in place and leaves them there; else Y is its scratch -- and was no log before)
            const int scorer_src = !fuse_logs ? 0 : ((use_em && !sp.no_scorer_reuse) ? 1 : 2);
            const int dr = UNI(lds_decimate<FORCE>(lds_offset_of(smem), lds_offset_of(&s_cold), nt, n, m, ne, cur, active, pi, simplified, scorer_src));
            decimated = dr & DEC_FIXED; simplified = (dr >> DEC_VERIFIED_SHIFT) & 3;
            const int spec_bits = (dr >> DEC_SPEC_SHIFT) & 3;
            used |= 4u;
            if (spec_bits & 1) zero |= 4u;
            if (spec_bits & 2) violation = 1;
        }
        if (decimated) { if (simplified == 1) PROF_COUNT(10); else PROF_COUNT(11); }
        PROF_COUNT(12);
        PROF_MARK(6);                                        // P6 decimation
        if (has_prev) cnt = cnt + 1.0f;
        // the speculation record of this iteration: kept in LDS and flushed once at the end of the launch (a global atomic here would
        // make thread 0's wave wait for an L2 round trip at the next barrier, in every iteration)
        if (tid == 0 && !poisoned) {
            if (t < SPEC_LOCAL) { s_spec_used[t] = (uint8_t)used; s_spec_zero[t] = (uint8_t)zero; }
            else { atomicOr(&sp.spec_used[t], used); if (zero) atomicOr(&sp.spec_zero[t], zero); }
        }
        // ---- P7: edge-mask refresh (only changes after a decimation) and P8: termination check --------------------------------
        // (trainer.py:150-162 through SatCNFEvaluator, util.py:226-236: the count of satisfied clauses only changes when `sol` does;
        //  the two passes share one barrier, and the count's reduction is one DPP sum per wave + one ds_add per wave)
        {
            const bool refresh = decimated || !use_em;
            const bool recount = sp.check_termination && (decimated || rf_changed || nsat < 0);
            if (refresh) {
                const float *const av = L.av, *const af = L.af;
                for (int p = tid; p < ne; p += nt) {
                    const uint16_t cw = pcc[p];
                    const float a = 0.0f + av[pvv[p] & VM];
                    const float b = 0.0f + af[cw & 0x3fff];
                    const bool em = (a * b) == 1.0f;
                    pcc[p] = (uint16_t)((cw & ~PC_EM) | (em ? PC_EM : 0));
                }
                use_em = 1; em_dirty = 1; mask_fix = logs_ready;
                PROF_COUNT(17);
            }
            PROF_MARK(14);                                   // P7
            if (recount) {
                const uint16_t *const e2p = L.e2p, *const f_ptr = L.f_ptr;
                const float *const sol = L.sol;
                auto lit = [&](int k) {                      // literal value (s x + (1 - s) / 2) > 0.5, util.py:228-231
                    const uint16_t pw = pvv[e2p[k]];
                    const float sg = slot_sign(pw);
                    float ev = 0.0f + sg * sol[pw & VM];
                    ev = ev + (1.0f - sg) / 2.0f;
                    return (ev > 0.5f) ? 1.0f : 0.0f;
                };
                int cs = 0;
                for (int c0 = tid; c0 < m; c0 += 2 * nt) {    // two clauses per trip: the three-level LDS chains of both in flight together
                    const int c1 = c0 + nt;
                    const bool has1 = c1 < m;
                    const int d1 = has1 ? c1 : c0;
                    const int a0 = f_ptr[c0], b0 = f_ptr[c0 + 1], a1 = f_ptr[d1], b1 = f_ptr[d1 + 1];
                    if (__builtin_amdgcn_ballot_w64(b0 - a0 != 3 || b1 - a1 != 3) == 0) {
                        const float t0 = lit(a0), t1 = lit(a0 + 1), t2 = lit(a0 + 2), u0 = lit(a1), u1 = lit(a1 + 1), u2 = lit(a1 + 2);
                        float cl0 = 0.0f, cl1 = 0.0f;
                        cl0 = cl0 + t0; cl0 = cl0 + t1; cl0 = cl0 + t2; cl1 = cl1 + u0; cl1 = cl1 + u1; cl1 = cl1 + u2;
                        cs += ((cl0 > 0.0f) ? 1 : 0) + ((has1 && cl1 > 0.0f) ? 1 : 0);
                    } else {
                        float cl0 = 0.0f, cl1 = 0.0f;
                        for (int k = a0; k < b0; ++k) cl0 = cl0 + lit(k);
                        if (has1) for (int k = a1; k < b1; ++k) cl1 = cl1 + lit(k);
                        cs += ((cl0 > 0.0f) ? 1 : 0) + ((has1 && cl1 > 0.0f) ? 1 : 0);
                    }
                }
                cs = wave_reduce(cs, OpAddI(), 0);
                if (lane == 63 && cs) atomicAdd(&s_sat_count, cs);
                PROF_COUNT(16);
            }
            PROF_MARK(18);                                   // P8: the count itself (its barrier is in mark 15)
            if (refresh || recount) __syncthreads();
            if (recount) nsat = UNI(s_sat_count);            // (thread 0 clears the word behind the next sweep's R1 barrier)
            if (sp.check_termination && active && nsat == m) active = 0;
        }
        has_prev = 1; prev_from_global = 0; cur ^= 1;
        PROF_MARK(15);                                       // P8
    }

    // ---- leave: either the instance is finished (inactive, or the loop ends here) and its results go to the caller's arrays,
    // or it continues in the next launch and its LDS image goes to the next dynamic record -----------------------------------
    int any_inactive = 0;
    for (int v = tid; v < n; v += nt) any_inactive |= (L.av[v] == 0.0f) ? 1 : 0;
    any_inactive = __syncthreads_or(any_inactive);
    const bool finishing = did_prop && (sp.final_chunk || !active);
    float *const Efin = cur ? L.EB : L.EA, *const Eprev = cur ? L.EA : L.EB;   // after the toggle: final surveys / the ones the last sweep read
    // left inactive with iterations still to come: its frozen state is looked at right here (lds_ghost_bad)
    const bool ghost = finishing && !active && !(sp.final_chunk && iters >= T) && !sp.isolate;
    if (finishing) {
        float *gq = sp.q + 3 * (size_t)G.e0;
        float *gfs = sp.fs + 2 * (size_t)G.e0;
        for (int p = tid; p < ne; p += nt) {
            const int e = G.v_edges[p];
            const uint16_t pw = pvv[p], cw = pcc[p];
            const int v = pw & VM;
            const float s = slot_sign(pw);
            float y = pdp_safe_log_fin(1.0f - Eprev[p], PDP_SP_EPS);
            if (last_use_em) y = y * ((cw & PC_EM_USED) ? 1.0f : 0.0f);
            const float force = RF ? (rf_last_flip ? (qu_is_delta ? QU[p] : X[p]) : frc_of(pw)) : (FORCE ? frc_of(pw) : 0.0f);
            const float pos = 0.0f + L.Pv[v], neg = 0.0f + L.Nv[v];
            float same = (0.5f * (1.0f + s)) * pos + (0.5f * (1.0f - s)) * neg;
            same = same - y;
            same = same + ((force == s) ? L1 : L0);
            float opp = (0.5f * (1.0f - s)) * pos + (0.5f * (1.0f + s)) * neg;
            opp = opp + ((force == -s) ? L1 : L0);
            const float dc = pdp_expf_fin_le30(same + opp);
            const float A = pdp_expf_fin_le30(same), Bv = pdp_expf_fin_le30(opp);
            const float qu = A * (1.0f - Bv), qs = Bv * (1.0f - A);
            const float total = (qu + qs) + dc;
            // (1 - mask) * old keeps a NaN forever; the three columns of q turn NaN together, so q_u carries the stickiness
            // q_u: stored by a sweep of the plain form; after a sweep that took the logs, formed as that sweep formed it ((+0) * the log of the
            // value keeps a NaN: X holds log(max(q_u, eps)) * mask of exactly that value)
            const float sticky = qu_is_delta ? __builtin_fmaf(1.0f - 1.0f, X[p], qu / total) : QU[p];
            // (PDP_DEBUG_GHOST_INJECT=1, tests only: the first slot of a leaving instance gets a NaN q_u, so that the ghost sweep has something to find)
            const float sticky_out = (ghost && sp.debug_ghost_inject && p == 0) ? __builtin_nanf("") : sticky;
            if (ghost) QU[p] = sticky_out;                  // (lds_ghost_bad reads the q_u the caller gets)
            gq[3 * e] = sticky_out;
            gq[3 * e + 1] = 1.0f * (qs / total) + (1.0f - 1.0f) * sticky;
            gq[3 * e + 2] = 1.0f * (dc / total) + (1.0f - 1.0f) * sticky;
            gfs[2 * e] = Efin[p];
            if constexpr (RF) gfs[2 * e + 1] = frc_of(pw);
            sp.prev[G.e0 + e] = Efin[p];
            G.emask[e] = (cw & PC_EM) ? 1.0f : 0.0f;
        }
        for (int v = tid; v < n; v += nt) { G.av[v] = L.av[v]; G.sol[v] = L.sol[v]; }
        for (int c = tid; c < m; c += nt) G.af[c] = L.af[c];
    } else {
        auto dump16 = [&](size_t off, const void *src, size_t bytes) {
            uint4 *g = reinterpret_cast<uint4 *>(dout + off);
            const uint4 *d = reinterpret_cast<const uint4 *>(src);
            for (int i = tid; i < (int)((bytes + 15) >> 4); i += nt) g[i] = d[i];
        };
        dump16(BL.QU, QU, (size_t)ne * 4); dump16(BL.E, Efin, (size_t)ne * 4);
        dump16(BL.af, L.af, (size_t)m * 4); dump16(BL.av, L.av, (size_t)n * 4); dump16(BL.sol, L.sol, (size_t)n * 4);
        // the next launch starts with "the mask the last propagate used" == the current mask
        uint16_t *gpc = reinterpret_cast<uint16_t *>(dout + BL.pcc);
        for (int p = tid; p < ne; p += nt) { const uint16_t cw = pcc[p]; gpc[p] = (uint16_t)((cw & ~PC_EM_USED) | ((cw & PC_EM) ? PC_EM_USED : 0)); }
        if constexpr (RF) { for (int p = tid; p < ne; p += nt) sp.frc_out[G.e0 + p] = frc_of(pvv[p]); }
    }
    int ghost_bad = 0;
    if (ghost) {
        __syncthreads();
        ghost_bad = UNI(lds_ghost_bad(lds_offset_of(smem), lds_offset_of(&s_cold), nt, n, m, ne, cur ? 1 : 0, pi, (FORCE || RF) ? 1 : 0, RF ? 0 : 1, VM));
    }
    if (tid < SPEC_LOCAL && tid < T) {
        if (s_spec_used[tid]) atomicOr(&sp.spec_used[tid], (uint32_t)s_spec_used[tid]);
        if (s_spec_zero[tid]) atomicOr(&sp.spec_zero[tid], (uint32_t)s_spec_zero[tid]);
    }
    PROF_MARK(8);                                            // write back
    PROF_FLUSH();
    if (tid == 0) {
        DynHeader h;
        h.active = (uint32_t)active; h.done = finishing ? 1u : 0u; h.perm_zero = (finishing && any_inactive) ? 1u : 0u; h.simplified = (uint32_t)simplified;
        h.cnt = cnt; h.is_sat = s_cold.is_sat; h.nsat_p1 = nsat + 1; h.pad2 = 0.0f;
        *reinterpret_cast<DynHeader *>(dout + BL.hdr) = h;
        const int gb = (LISTED || REPLAY) ? *(volatile int *)&s_inst : G.b;
        if (finishing) { sp.amask[gb] = (uint8_t)active; sp.counters[gb] = cnt; pv_.is_sat[gb] = s_cold.is_sat; }
        // the verdict of the ghost sweep (2: something non-finite: the call fails, k_solve_finish of the last chunk collects the flags)
        if (ghost) sp.ghost_flag[gb] = ghost_bad ? 2 : 0;
        else if (REPLAY) sp.ghost_flag[gb] = 0;           // what pass 1 did with this instance behind the poison is void, its flag with it
        if (any_inactive) atomicMin(&ctl->perm_zero, (uint32_t)iters);
        if (!REPLAY) sp.last_event[gb] = last_event;
        if (sp.risk) sp.risk[gb] = finishing ? 0xffffffffu : s_risk;
        atomicMax(&ctl->iters_run, (uint32_t)iters);
        if (violation) atomicOr(&ctl->violation, 1u);
    }
}

// canonical arrays -> static + first dynamic record (once per call; the static part once per problem)
// call-entry state of a pdp_sp_solve call (restored when its speculation fails)
struct SolveSnapshot {
    float *q, *fs, *av, *af, *sol, *sat, *emask, *prev, *cnt; uint8_t *amask;
};
__global__ void __launch_bounds__(256) k_solve_import(PView pv, const float *q, const float *fs, const uint8_t *amask, const float *prev, const float *counters,
                                                      int has_prev, int build_static, char *stat, char *dyn, const int64_t *stat_off, const int64_t *dyn_off, float *prev_slots,
                                                      const int32_t *list, int stage_cap /* slots per column of the LDS staging area */,
                                                      SolveCall *force_check /* non-NULL: raise force_seen when fs[:, 1] holds anything but zeros */,
                                                      int use_em /* 0: the problem has no edge mask yet (all ones) */,
                                                      SolveSnapshot snap /* .av non-NULL: also take the call-entry snapshot of what is read here anyway */)
{
    const Inst G = load_inst(pv, list ? list[blockIdx.x] : (int)blockIdx.x);
    const int n = G.n, m = G.m, ne = G.e, tid = threadIdx.x, nt = blockDim.x;
    const BlobLayout BL = blob_layout(n, m, ne);
    char *st = stat + stat_off[G.b], *dy = dyn + dyn_off[G.b];
    uint16_t *pvv = reinterpret_cast<uint16_t *>(st + BL.pvv), *e2p = reinterpret_cast<uint16_t *>(st + BL.e2p);
    float *QU = reinterpret_cast<float *>(dy + BL.QU), *Ecur = reinterpret_cast<float *>(dy + BL.E);
    uint16_t *pcc = reinterpret_cast<uint16_t *>(dy + BL.pcc);
    const float *sq = q + 3 * (size_t)G.e0, *sfs = fs + 2 * (size_t)G.e0;
    // Slots are variable-major, edge ids clause-major: a slot's sources are scattered over the instance's edge range, 64 cache lines per
    // wave load.  When the instance's columns fit the staging area they are read in EDGE order (a wave load spans 4-12 lines), parked in
    // LDS and gathered from there; records are written in slot order either way.  (solve call 12.81 -> 12.67 ms on the headline batch.)
    extern __shared__ __attribute__((aligned(16))) float stage[];
    if (force_check) {
        // the force column shares its cache lines with the surveys read below: no extra HBM traffic (a NaN counts as a force)
        // (four edges per trip: independent loads, one round trip for the four)
        int some = 0;
        for (int e0 = tid; e0 < ne; e0 += 4 * nt) {
            float f[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) { const int e = e0 + j * nt; f[j] = sfs[2 * (e < ne ? e : e0) + 1]; }
#pragma unroll
            for (int j = 0; j < 4; ++j) some |= (f[j] != 0.0f) ? 1 : 0;
        }
        if (__syncthreads_or(some) && tid == 0) atomicOr(&force_check->force_seen, 1u);
    }
    if (ne <= stage_cap) {
        // staged per edge: q_u, the survey, the slot's clause word with its mask bits (16 bits) and -- only when the decimator has one -- the
        // previous survey: 10 or 14 bytes per edge, so that six (four) workgroups share a CU where 20 bytes let three
        float *sQ = stage, *sE = stage + stage_cap, *sP = stage + 2 * stage_cap;
        uint16_t *sC = reinterpret_cast<uint16_t *>(stage + (has_prev ? 3 : 2) * stage_cap);
        // (both loops take several elements per trip -- all their global loads first, then the LDS traffic: a one-element loop is one round
        //  trip per element, ten in a row per loop for n = 200 on 256 threads)
        for (int e0 = tid; e0 < ne; e0 += 4 * nt) {
            float qv[4], ev[4], mv[4] = {1.0f, 1.0f, 1.0f, 1.0f}, pv[4] = {0.0f, 0.0f, 0.0f, 0.0f}; int fn[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int e = (e0 + j * nt < ne) ? e0 + j * nt : e0;
                qv[j] = sq[3 * e]; ev[j] = sfs[2 * e]; fn[j] = G.e_fn[e];
                if (use_em) mv[j] = G.emask[e];                       // (no edge mask yet: all ones, pdp_problem_bind_state)
                if (has_prev) pv[j] = prev[G.e0 + e];
            }
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int e = e0 + j * nt;
                if (e >= ne) continue;
                sQ[e] = qv[j]; sE[e] = ev[j]; sC[e] = (uint16_t)(fn[j] | ((mv[j] == 1.0f) ? (PC_EM | PC_EM_USED) : 0));
                if (has_prev) sP[e] = pv[j];
            }
        }
        __syncthreads();
        for (int p0 = tid; p0 < ne; p0 += 8 * nt) {
            int e[8], evar[8], esg[8];
#pragma unroll
            for (int j = 0; j < 8; ++j) { const int p = p0 + j * nt; e[j] = G.v_edges[p < ne ? p : p0]; }
            if (build_static) {
#pragma unroll
                for (int j = 0; j < 8; ++j) { evar[j] = G.e_var[e[j]]; esg[j] = G.sgn[e[j]]; }
            }
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const int p = p0 + j * nt;
                if (p >= ne) continue;
                if (build_static) { pvv[p] = (uint16_t)(evar[j] | (esg[j] < 0 ? 0x8000 : 0)); e2p[e[j]] = (uint16_t)p; }
                pcc[p] = sC[e[j]];
                QU[p] = sQ[e[j]]; Ecur[p] = sE[e[j]];
                if (has_prev) prev_slots[G.e0 + p] = sP[e[j]];
            }
        }
    } else
    // four slots per trip: the gathers through v_edges are dependent loads, keep several of them in flight
    for (int p0 = tid; p0 < ne; p0 += 4 * nt) {
        int e[4]; float qv[4], ev[4], mv[4], pv[4] = {0.0f, 0.0f, 0.0f, 0.0f}; int fn[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) { const int p = p0 + j * nt; e[j] = G.v_edges[p < ne ? p : p0]; }
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            qv[j] = sq[3 * e[j]]; ev[j] = sfs[2 * e[j]]; mv[j] = use_em ? G.emask[e[j]] : 1.0f; fn[j] = G.e_fn[e[j]];
            if (has_prev) pv[j] = prev[G.e0 + e[j]];
        }
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int p = p0 + j * nt;
            if (p >= ne) continue;
            if (build_static) { pvv[p] = (uint16_t)(G.e_var[e[j]] | (G.sgn[e[j]] < 0 ? 0x8000 : 0)); e2p[e[j]] = (uint16_t)p; }
            pcc[p] = (uint16_t)(fn[j] | ((mv[j] == 1.0f) ? (PC_EM | PC_EM_USED) : 0));
            QU[p] = qv[j]; Ecur[p] = ev[j];
            if (has_prev) prev_slots[G.e0 + p] = pv[j];
        }
    }
    if (build_static) {
        uint16_t *vp = reinterpret_cast<uint16_t *>(st + BL.vptr), *fp = reinterpret_cast<uint16_t *>(st + BL.fptr);
        for (int v = tid; v <= n; v += nt) vp[v] = (uint16_t)G.v_ptr[v];
        for (int c = tid; c <= m; c += nt) fp[c] = (uint16_t)G.f_ptr[c];
        // counting sort of the variables by descending degree (capped at 255); the order inside a bucket is irrelevant
        __shared__ int hist[256];
        uint16_t *vord = reinterpret_cast<uint16_t *>(st + BL.vord);
        for (int i = tid; i < 256; i += nt) hist[i] = 0;
        __syncthreads();
        for (int v = tid; v < n; v += nt) { const int d = G.v_ptr[v + 1] - G.v_ptr[v]; atomicAdd(&hist[255 - (d < 255 ? d : 255)], 1); }
        __syncthreads();
        if (tid == 0) { int run = 0; for (int i = 0; i < 256; ++i) { const int c = hist[i]; hist[i] = run; run += c; } }
        __syncthreads();
        for (int v = tid; v < n; v += nt) { const int d = G.v_ptr[v + 1] - G.v_ptr[v]; vord[atomicAdd(&hist[255 - (d < 255 ? d : 255)], 1)] = (uint16_t)v; }
    }
    float *av = reinterpret_cast<float *>(dy + BL.av), *sol = reinterpret_cast<float *>(dy + BL.sol), *af = reinterpret_cast<float *>(dy + BL.af);
    int any_inactive = 0;
    for (int v = tid; v < n; v += nt) {
        const float a = G.av[v], so = G.sol[v];
        av[v] = a; sol[v] = so; any_inactive |= (a == 0.0f) ? 1 : 0;
        if (snap.av) { snap.av[G.v0 + v] = a; snap.sol[G.v0 + v] = so; }
    }
    for (int c = tid; c < m; c += nt) { const float f = G.af[c]; af[c] = f; if (snap.av) snap.af[G.f0 + c] = f; }
    any_inactive = __syncthreads_or(any_inactive);
    if (tid == 0) {
        if (snap.av) { snap.sat[G.b] = pv.is_sat[G.b]; snap.cnt[G.b] = counters[G.b]; snap.amask[G.b] = amask[G.b]; }
        DynHeader h;
        h.active = amask[G.b] ? 1u : 0u; h.done = h.active ? 0u : 1u;      // an instance that enters inactive never runs: nothing to write back
        h.perm_zero = (h.done && any_inactive) ? 1u : 0u; h.simplified = 0;
        h.cnt = counters[G.b]; h.is_sat = pv.is_sat[G.b]; h.nsat_p1 = 0; h.pad2 = 0.0f;
        *reinterpret_cast<DynHeader *>(dy + BL.hdr) = h;
    }
}

// the force column of the caller's [E,2] state in slot order (Reinforce; the SP triple with an external force)
__global__ void __launch_bounds__(256) k_force_import(PView pv, const float *fs, float *frc, SolveCtl *ctl0, const uint8_t *amask)
{
    const Inst G = load_inst(pv, blockIdx.x);
    if (!amask[G.b]) return;                      // an instance that enters inactive never runs: its force is never read
    const float *sfs = fs + 2 * (size_t)G.e0;
    int odd = 0;
    for (int p = threadIdx.x; p < G.e; p += blockDim.x) {
        const float f = sfs[2 * G.v_edges[p] + 1];
        frc[G.e0 + p] = f;
        odd |= (f == f && f != 0.0f && f != 1.0f && f != -1.0f) ? 1 : 0;     // the resident kernel keeps the force as a 2-bit code
    }
    if (__syncthreads_or(odd) && threadIdx.x == 0) atomicOr(&ctl0->violation, 1u);   // the call fails over to the step-wise loop
}

// after pass 1 of a chunk: does a NaN poison the batch from this chunk on?  
// Between two solver launches of a chunk (one workgroup each: every launch less is ~5 us of stream time less, four times per call).
// After pass 1: is the batch poisoned in this chunk, and which instances does the poison replay take?  Only instances with a gate /
// convergence event at or after the poison iteration behave differently under the poison (big instances: replayed wholesale).
__global__ void __launch_bounds__(1024) k_solve_post(SolveCtl *ctl, SolveCall *call, int c, int isolate, int B, const int32_t *last_event, int32_t *list,
                                                     const uint8_t *is_big)
{
    __shared__ int s_from;
    if (threadIdx.x == 0) {
        int from = -1;
        if (!call->stop) {
            ctl->do_replay = 0; ctl->replay_count = 0;
            // (isolated instances: a NaN stays inside its instance, nothing is replayed)
            if (!isolate && !call->poisoned_all && ctl->nan_iter < (uint32_t)c) {
                from = (int32_t)ctl->nan_iter;
                ctl->poison_from = from;
                call->poisoned_all = 1;
                if (ctl->violation) call->fail = 1;
                ctl->do_replay = 1;
            }
        }
        s_from = from;
    }
    __syncthreads();
    const int from = s_from;
    if (from < 0) return;
    for (int b = threadIdx.x; b < B; b += blockDim.x)
        if (!(is_big && is_big[b]) && last_event[b] >= from) list[atomicAdd(&ctl->replay_count, 1u)] = b;
}

// After the (possible) replay: speculation check of the chunk, loop control and -- `order` given -- the dispatch order of the next chunk's pass 1.
// Dispatch order of the next launch's pass 1: ascending by the binary exponent of the smallest q normalisation (q_u + q_s + q_dc) an instance
// met in its last sweep.  The batch's first NaN survey is a 0 / 0 of that normalisation (pdp_propagate.py:215-216): both products of a variable
// underflow, and they get there over several sweeps -- so the instances closest to it start in the first round of workgroups, put the NaN sweep
// on record early, and the workgroups of the later rounds take it as their poison point at once instead of being replayed (k_sp_solve_lds:
// `adopt`).  Only the ORDER depends on this guess: an instance's result does not depend on when it runs.  One workgroup: a counting sort over
// the 256 exponents (finished instances and NaNs carry all ones: last), the order inside a bucket is whatever the atomics make it.
__global__ void __launch_bounds__(1024) k_solve_finish(SolveCtl *ctl, SolveCall *call, const uint32_t *spec_used, const uint32_t *spec_zero, int c, int chunk_start,
                                                       int isolate, int B, const uint32_t *risk, int32_t *order, const uint8_t *ghost_flag)
{
    __shared__ int hist[256];
    __shared__ int s_stop;
    const int tid = threadIdx.x, nt = blockDim.x;
    if (ghost_flag) {
        // (the call's last chunk) an instance that left inactive found its frozen state non-finite under the reference's masked sweep: the
        // speculation "an inactive instance changes nothing" does not hold, the call fails over
        int bad = 0;
        for (int b = tid; b < B; b += nt) bad |= (ghost_flag[b] == 2) ? 1 : 0;
        if (bad) call->fail = 1;
    }
    if (tid == 0) s_stop = call->stop ? 1 : 0;
    for (int i = tid; i < 256; i += nt) hist[i] = 0;
    __syncthreads();
    if (s_stop) return;
    // iterations at or after the poison point are not speculated on (the reference's reductions are NaN there), and the
    // kernel records nothing in a poisoned iteration
    const int poison_from = ctl->do_replay ? ctl->poison_from : 0x7fffffff;
    const uint32_t perm_zero = ctl->perm_zero;
    if (!isolate) {
        int bad = 0;
        for (int t = tid; t < c && t < poison_from; t += nt)
            if ((uint32_t)t < perm_zero && (spec_used[t] & ~spec_zero[t]) != 0u) bad = 1;
        if (bad) call->fail = 1;
    }
    __syncthreads();                                   // (everybody has read s_stop)
    if (tid == 0) {
        if (ctl->violation && !isolate) call->fail = 1;
        const uint32_t it = ctl->iters_run;
        call->total_iters = (uint32_t)chunk_start + it;
        const int stop = it < (uint32_t)c ? 1 : 0;     // every instance went inactive inside this chunk (global early exit, solver.py:383)
        if (stop) call->stop = 1;
        s_stop = stop;
    }
    if (!order) return;
    __syncthreads();
    if (s_stop) return;
    for (int b = tid; b < B; b += nt) atomicAdd(&hist[(risk[b] >> 23) & 255], 1);
    __syncthreads();
    if (tid == 0) { int run = 0; for (int i = 0; i < 256; ++i) { const int h = hist[i]; hist[i] = run; run += h; } }
    __syncthreads();
    for (int b = tid; b < B; b += nt) order[atomicAdd(&hist[(risk[b] >> 23) & 255], 1)] = b;
}


// chunk-entry state of the big instances (HBM-resident kernel works in place): save before pass 1, restore before the replay pass
struct BigSnap { float *q, *fs, *av, *af, *sol, *sat, *emask, *prev, *cnt; uint8_t *amask; };
__global__ void __launch_bounds__(256) k_big_state(PView pv, const int32_t *list, BigSnap live, BigSnap snap, int restore, const SolveCtl *ctl, const SolveCall *call)
{
    if (call->stop) return;
    if (restore && !ctl->do_replay) return;
    const Inst G = load_inst(pv, list[blockIdx.x]);
    const BigSnap &src = restore ? snap : live, &dst = restore ? live : snap;
    const int tid = blockIdx.y * blockDim.x + threadIdx.x, nt = gridDim.y * blockDim.x;
    for (int64_t i = tid; i < 3 * (int64_t)G.e; i += nt) dst.q[3 * (int64_t)G.e0 + i] = src.q[3 * (int64_t)G.e0 + i];
    for (int64_t i = tid; i < 2 * (int64_t)G.e; i += nt) dst.fs[2 * (int64_t)G.e0 + i] = src.fs[2 * (int64_t)G.e0 + i];
    for (int i = tid; i < G.e; i += nt) { dst.emask[G.e0 + i] = src.emask[G.e0 + i]; dst.prev[G.e0 + i] = src.prev[G.e0 + i]; }
    for (int i = tid; i < G.n; i += nt) { dst.av[G.v0 + i] = src.av[G.v0 + i]; dst.sol[G.v0 + i] = src.sol[G.v0 + i]; }
    for (int i = tid; i < G.m; i += nt) dst.af[G.f0 + i] = src.af[G.f0 + i];
    if (tid == 0) { dst.sat[G.b] = src.sat[G.b]; dst.cnt[G.b] = src.cnt[G.b]; dst.amask[G.b] = src.amask[G.b]; }
}


// ---- simplify() of a batch whose instances fit the LDS ----------------------------------------------------------------------------------
// pdp_simplify's per-instance kernel walks the HBM-resident arrays: a few latency-bound passes of dependent gathers per instance (252 us
// on the headline batch, a fiftieth of a 100-sweep solve).  Here the workgroup first copies the instance's topology into LDS in the
// variable-major slot form of the persistent solver (packed 16-bit words) and runs the SAME routines (d_simplify on an LView) there.
static size_t simplify_lds_bytes(int n, int m, int e)
{
    auto a16 = [](size_t x) { return (x + 15) & ~(size_t)15; };
    return 3 * a16((size_t)e * 2) + a16((size_t)(n + 1) * 2) + a16((size_t)(m + 1) * 2) + 3 * a16((size_t)n * 4) + a16((size_t)m * 4) + 2 * a16((size_t)n * 4) +
           a16((size_t)n) + 2 * a16((size_t)m);
}
// TOPO 0: the slot form is gathered from the problem's CSR arrays (two levels of dependent loads per slot: five sixths of this kernel's time on
// the headline batch); 1: gathered and left in `topo` (the second simplify() of a problem); 2: read back from there, five coalesced streams of
// 16-bit words (a problem whose state is re-bound -- pdp_problem_bind_state -- keeps its topology).
struct SimpTopo { uint16_t *pvv, *pcc, *e2p, *vptr, *fptr; };
template <int TOPO>
__global__ void __launch_bounds__(256) k_simplify_lds(PView pv, SimpTopo topo)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    __shared__ int redi[PDP_RED_SMALL];
    const Inst G = load_inst(pv, blockIdx.x);
    const int n = G.n, m = G.m, ne = G.e, tid = threadIdx.x, nt = blockDim.x;
    unsigned char *cp = smem;
    uint16_t *pvv = carve<uint16_t>(cp, ne), *pcc = carve<uint16_t>(cp, ne), *e2p = carve<uint16_t>(cp, ne);
    uint16_t *v_ptr = carve<uint16_t>(cp, n + 1), *f_ptr = carve<uint16_t>(cp, m + 1);
    float *av = carve<float>(cp, n), *sol = carve<float>(cp, n), *assign = carve<float>(cp, n), *af = carve<float>(cp, m);
    int32_t *deg = carve<int32_t>(cp, n), *sdeg = carve<int32_t>(cp, n);
    uint8_t *flag_v = carve<uint8_t>(cp, n), *flag_f = carve<uint8_t>(cp, m), *flag_f2 = carve<uint8_t>(cp, m);
    // (an instance's part of every kept array starts at an even element: 4-byte aligned for the LDS-DMA below)
    const size_t tb_e = ((size_t)G.e0 + 2 * (size_t)G.b) & ~(size_t)1;
    const size_t tb_v = ((size_t)G.v0 + 3 * (size_t)G.b) & ~(size_t)1, tb_f = ((size_t)G.f0 + 3 * (size_t)G.b) & ~(size_t)1;
    uint16_t *const t_pvv = topo.pvv + tb_e, *const t_pcc = topo.pcc + tb_e, *const t_e2p = topo.e2p + tb_e;
    uint16_t *const t_vptr = topo.vptr + tb_v, *const t_fptr = topo.fptr + tb_f;
    if constexpr (TOPO == 2) {
        // everything the kernel reads lands in LDS by DMA (global_load_lds_dword: no register round trip, every request of the workgroup in
        // flight at once, one wait in front of the barrier); the copy loops this replaces were a dependent round trip each
        const int lane = tid & 63;
        auto dma4 = [&](void *dst, const void *src, size_t bytes) {
            const int n4 = (int)((bytes + 3) >> 2);
            for (int i = tid; i < n4; i += nt)
                __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)(reinterpret_cast<const char *>(src) + (size_t)i * 4),
                                                 (__attribute__((address_space(3))) void *)(reinterpret_cast<uint32_t *>(dst) + (i - lane)), 4, 0, 0);
        };
        dma4(pvv, t_pvv, (size_t)ne * 2); dma4(pcc, t_pcc, (size_t)ne * 2); dma4(e2p, t_e2p, (size_t)ne * 2);
        dma4(v_ptr, t_vptr, (size_t)(n + 1) * 2); dma4(f_ptr, t_fptr, (size_t)(m + 1) * 2);
        dma4(av, G.av, (size_t)n * 4); dma4(sol, G.sol, (size_t)n * 4); dma4(af, G.af, (size_t)m * 4);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    } else {
    // (eight slots per trip: the gathers through v_edges are two levels of dependent loads, and a loop that takes one slot per trip is one
    //  pair of round trips per slot -- twenty in a row for n = 200 on 256 threads, which is what this kernel's time was made of)
    for (int p0 = tid; p0 < ne; p0 += 8 * nt) {
        int e[8], ev[8], ef[8], sg[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) { const int p = p0 + j * nt; e[j] = G.v_edges[p < ne ? p : p0]; }
#pragma unroll
        for (int j = 0; j < 8; ++j) { ev[j] = G.e_var[e[j]]; sg[j] = G.sgn[e[j]]; ef[j] = G.e_fn[e[j]]; }
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const int p = p0 + j * nt;
            if (p >= ne) continue;
            pvv[p] = (uint16_t)(ev[j] | (sg[j] < 0 ? 0x8000 : 0));
            pcc[p] = (uint16_t)ef[j];
            e2p[e[j]] = (uint16_t)p;                    // edges are clause-major: edge id == position in the by-clause list
        }
    }
    for (int v = tid; v <= n; v += nt) v_ptr[v] = (uint16_t)G.v_ptr[v];
    for (int c = tid; c <= m; c += nt) f_ptr[c] = (uint16_t)G.f_ptr[c];
    for (int v = tid; v < n; v += nt) { av[v] = G.av[v]; sol[v] = G.sol[v]; }
    for (int c = tid; c < m; c += nt) af[c] = G.af[c];
    }
    __syncthreads();
    if constexpr (TOPO == 1) {
        for (int p = tid; p < ne; p += nt) { t_pvv[p] = pvv[p]; t_pcc[p] = pcc[p]; t_e2p[p] = e2p[p]; }
        for (int v = tid; v <= n; v += nt) t_vptr[v] = v_ptr[v];
        for (int c = tid; c <= m; c += nt) t_fptr[c] = f_ptr[c];
    }
    LView I;
    I.b = G.b; I.n = n; I.m = m; I.e = ne; I.nt = nt; I.red = redi;
    I.e_var.pv = pvv; I.e_var.mask = 0x3fff; I.e_fn.pc = pcc; I.sgn.pv = pvv; I.f_edges = e2p; I.v_ptr = v_ptr; I.f_ptr = f_ptr;
    I.av = av; I.af = af; I.sol = sol;
    SimplifyScratch ss;
    ss.assign = assign; ss.deg = deg; ss.sdeg = sdeg; ss.flag_v = flag_v; ss.flag_f = flag_f; ss.flag_f2 = flag_f2; ss.red = redi;
    d_simplify(I, ss, pv.is_sat + G.b);
    for (int v = tid; v < n; v += nt) { G.av[v] = av[v]; G.sol[v] = sol[v]; }
    for (int c = tid; c < m; c += nt) G.af[c] = af[c];
}

// returns 1 if the LDS-resident form took the batch
int pdp_simplify_lds(pdp_problem *p, hipStream_t st)
{
    if (!p->fn_edges_identity || p->max_n >= 16384 || p->max_m >= 16384 || p->max_e >= 65536 || getenv("PDP_SIMPLIFY_HBM")) return 0;
    const size_t lds = simplify_lds_bytes(p->max_n, p->max_m, p->max_e);
    if (lds > 64 * 1024) return 0;
    // the slot topology is kept from the second call on (a problem that is simplified once -- the solver classes make a SATProblem per batch --
    // pays nothing for it)
    const size_t Bn = p->B, E = ((size_t)p->E + 2 * Bn + 8) & ~(size_t)7, rows_v = ((size_t)p->V + 3 * Bn + 8) & ~(size_t)7, rows_f = ((size_t)p->F + 3 * Bn + 8) & ~(size_t)7;
    int mode = 0;
    if (p->simp_calls >= 2 && p->simp_topo) mode = 2;
    else if (p->simp_calls == 1 && !getenv("PDP_SIMPLIFY_NO_TOPO") && pdp_dev_alloc((void **)&p->simp_topo, (3 * E + rows_v + rows_f + 8) * sizeof(uint16_t)) == PDP_OK) mode = 1;
    SimpTopo T;
    T.pvv = p->simp_topo; T.pcc = T.pvv + E; T.e2p = T.pcc + E; T.vptr = T.e2p + E; T.fptr = T.vptr + rows_v;
    if (!p->simp_topo) T.pvv = T.pcc = T.e2p = T.vptr = T.fptr = nullptr;
    const void *fn = mode == 2 ? (const void *)k_simplify_lds<2> : mode == 1 ? (const void *)k_simplify_lds<1> : (const void *)k_simplify_lds<0>;
    if (hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess) return 0;
    if (mode == 2) hipLaunchKernelGGL(k_simplify_lds<2>, dim3(p->B), dim3(256), lds, st, make_view(p), T);
    else if (mode == 1) hipLaunchKernelGGL(k_simplify_lds<1>, dim3(p->B), dim3(256), lds, st, make_view(p), T);
    else hipLaunchKernelGGL(k_simplify_lds<0>, dim3(p->B), dim3(256), lds, st, make_view(p), T);
    if (p->simp_calls < 2) p->simp_calls += 1;
    return 1;
}

// ---- host side ---------------------------------------------------------------------------------------------------

// skip: bit 0 the messages q / fs (the caller declared them disposable: pdp_solve_args.inputs_disposable), bit 1 the decimator's previous
// surveys (no previous state at call entry: the handle's has_prev flag says their content means nothing), bit 2 the edge mask (likewise
// p->has_edge_mask).  Only the call-entry snapshot of the LDS-resident path uses it: 378 MB of copies on config 2 shrink to 25 MB.
// up to ten device-to-device copies in ONE launch (ten hipMemcpyAsync were ten launches of the runtime's copy kernel, ~5 us each and serialised)
struct CopyList { const char *src[10]; char *dst[10]; size_t bytes[10]; int n; };
__global__ void __launch_bounds__(256) k_copy_list(CopyList c)
{
    const int k = blockIdx.y;
    if (k >= c.n) return;
    const size_t bytes = c.bytes[k], n16 = bytes >> 4;
    const char *src = c.src[k]; char *dst = c.dst[k];
    const bool aligned = ((((uintptr_t)src) | ((uintptr_t)dst)) & 15) == 0;
    const size_t start = blockIdx.x * (size_t)blockDim.x + threadIdx.x, stride = (size_t)gridDim.x * blockDim.x;
    if (aligned) {
        for (size_t i = start; i < n16; i += stride) reinterpret_cast<uint4 *>(dst)[i] = reinterpret_cast<const uint4 *>(src)[i];
        for (size_t i = (n16 << 4) + start; i < bytes; i += stride) dst[i] = src[i];
    } else {
        for (size_t i = start; i < bytes; i += stride) dst[i] = src[i];
    }
}
static int snapshot_copy(pdp_problem *p, pdp_solve_args *a, SolveSnapshot &s, bool save, hipStream_t st, int skip = 0)
{
    const size_t E = p->E, V = p->V, F = p->F, B = p->B;
    struct { void *live; void *snap; size_t bytes; int skip_bit; } items[] = {
        {a->q, s.q, 3 * E * 4, 1}, {a->fs, s.fs, 2 * E * 4, 1}, {p->av, s.av, V * 4, 0}, {p->af, s.af, F * 4, 0}, {p->sol, s.sol, V * 4, 0},
        {p->is_sat, s.sat, B * 4, 0}, {p->emask, s.emask, E * 4, 4}, {a->decimator->prev, s.prev, E * 4, 2},
        {a->decimator->counters, s.cnt, B * 4, 0}, {a->active_mask, s.amask, B, 0}};
    CopyList c; c.n = 0;
    size_t longest = 0;
    for (auto &it : items) {
        if ((skip & it.skip_bit) || it.bytes == 0) continue;
        c.src[c.n] = (const char *)(save ? it.live : it.snap); c.dst[c.n] = (char *)(save ? it.snap : it.live); c.bytes[c.n] = it.bytes;
        if (it.bytes > longest) longest = it.bytes;
        ++c.n;
    }
    if (c.n == 0) return PDP_OK;
    size_t gx = (longest / 16 + 255) / 256;
    if (gx < 1) gx = 1;
    if (gx > 4096) gx = 4096;
    hipLaunchKernelGGL(k_copy_list, dim3((unsigned)gx, (unsigned)c.n), dim3(256), 0, st, c);
    PDP_LAUNCH_CHECK();
    return PDP_OK;
}

__global__ void k_max_i32(int B, const int32_t *x, uint32_t *out)
{
    int m = 0;
    for (int b = blockIdx.x * blockDim.x + threadIdx.x; b < B; b += gridDim.x * blockDim.x) m = x[b] > m ? x[b] : m;
    atomicMax(out, (uint32_t)m);
}

// everything a call's device-side control starts from, in one launch: the per-chunk blocks, the call block, the speculation record, the
// ghost flags and the identity dispatch order (was: this kernel + two memsets + k_order_identity, four launches and their gaps)
__global__ void k_solve_ctl_init(SolveCtl *ctl, int nchunks, SolveCall *call, uint32_t *spec, int spec_words, uint8_t *ghost_flag, int B, int32_t *order, uint32_t *risk)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < nchunks) { SolveCtl c; memset(&c, 0, sizeof(c)); c.nan_iter = 0xffffffffu; c.perm_zero = 0xffffffffu; c.poison_from = 0x7fffffff; ctl[i] = c; }
    if (i == 0) { SolveCall z; memset(&z, 0, sizeof(z)); *call = z; }
    if (i < spec_words) spec[i] = 0u;
    if (i < B) { ghost_flag[i] = 0; if (order) { order[i] = i; risk[i] = 0xffffffffu; } }
}

static int ensure_bytes(char **ptr, size_t *have, size_t need)
{
    if (*have >= need) return PDP_OK;
    if (*ptr) pdp_dev_free(*ptr);
    *ptr = nullptr; *have = 0;
    { int st_ = pdp_dev_alloc((void **)ptr, need); if (st_ != PDP_OK) return st_; }
    *have = need;
    return PDP_OK;
}

// does this instance's image fit the LDS-resident solver?  (force-less image: the Reinforce force is a 2-bit code, an external-force
// column of the SP triple is checked by the caller against the largest fitting instance)
// Launch of the HBM-resident kernel over `count` instances.  Few instances get a TEAM of workgroups each: a team's workgroups wait
// for each other, so all of them must be resident at once, and a team is only as large as the instance has work for (>= 2 edges per
// thread).  Normally a team stays on one XCD (<= 32 workgroups, the teams of an XCD within one workgroup per CU): its barriers are
// cheap there.  `wide`: nothing else runs next to this launch (single-instance batches, batches of big instances only) and an
// instance is huge -- then a team may span the whole chip (<= 256 workgroups in all, one per CU) with agent-scope barriers, which
// cost ~10x more each and pay from a few hundred thousand edges on.  PDP_SOLVE_TEAM=<n> caps the team size (1: never a team),
// PDP_SOLVE_TEAM_WIDE_EDGES=<e> moves the threshold of the wide form (0: never).  (The plan itself: pdp_team_plan, pdp_problem.hip.)
static int launch_hbm(pdp_problem *p, SolveParams sp, int count, hipStream_t s_, bool wide = false)
{
    // Team workgroup size.  Next to the LDS-resident kernel (per-instance routing of a mixed batch): 256 -- measured on the mixed headline batch
    // (tools/mixed_batch_time.py), 256 x 32 beats 512 x 32 and 1024 x 16.  With the chip to itself (`wide`: exact single-instance mode, whole
    // batches of big instances) a team has one workgroup per CU, and 256 threads are one wave per SIMD with nothing to cover the gathers:
    // 1024 -- a forward of 100 sweeps at n = 20 000 / 100 000 / 300 000 / 1 000 000: 28 / 31 / 69 / 230 -> 18 / 26 / 52 / 154 ms.
    int tnt = wide ? 1024 : 256;
    if (const char *env = getenv("PDP_SOLVE_TEAM_THREADS")) { const int v = atoi(env); if (v == 256 || v == 512 || v == 1024) tnt = v; }
    TeamLaunch tl;
    { const int st_ = pdp_team_plan(p, count, wide, tnt, &tl, s_); if (st_ != PDP_OK) return st_; }
    if (tl.size > 1) {
        sp.team_size = tl.size; sp.team_count = tl.count; sp.team_slots = tl.slots; sp.team_ws = tl.ws; sp.team_no_xcd = tl.no_xcd;
        sp.spin_limit = tl.spin_limit;
        if (tnt == 1024) hipLaunchKernelGGL((k_sp_solve<1024, true>), dim3(tl.size * tl.slots), dim3(1024), 0, s_, make_view(p), sp);
        else if (tnt == 512) hipLaunchKernelGGL((k_sp_solve<512, true>), dim3(tl.size * tl.slots), dim3(512), 0, s_, make_view(p), sp);
        else hipLaunchKernelGGL((k_sp_solve<256, true>), dim3(tl.size * tl.slots), dim3(256), 0, s_, make_view(p), sp);
    } else if (count <= 512) {
        hipLaunchKernelGGL((k_sp_solve<1024, false>), dim3(count), dim3(1024), 0, s_, make_view(p), sp);     // one wave per SIMD waits on L2 most of the time
    } else {
        hipLaunchKernelGGL((k_sp_solve<256, false>), dim3(count), dim3(256), 0, s_, make_view(p), sp);
    }
    return PDP_OK;
}

static bool instance_fits_lds(int n, int m, int e)
{
    return lds2_bytes_for(n, m, e) <= 160 * 1024 - 1024 && e < 65535 && n < 16384 && m < 16384;
}

// once per problem: which instances fit the LDS (fit_list / big_list), the byte offsets of the fitting instances' records and the records'
// storage.  Instances that do not fit run on the HBM-resident kernel inside the same chunk loop (per-instance routing).
static int resident_prepare(pdp_problem *p)
{
    if (p->res_stat_off) return PDP_OK;
    const size_t B = p->B, E = p->E;
    std::vector<int32_t> v0(B + 1), f0(B + 1), e0(B + 1);
    PDP_HIP_CHECK(hipMemcpy(v0.data(), p->inst_v0, (B + 1) * 4, hipMemcpyDeviceToHost));
    PDP_HIP_CHECK(hipMemcpy(f0.data(), p->inst_f0, (B + 1) * 4, hipMemcpyDeviceToHost));
    PDP_HIP_CHECK(hipMemcpy(e0.data(), p->inst_e0, (B + 1) * 4, hipMemcpyDeviceToHost));
    std::vector<int64_t> off(2 * B);
    std::vector<int32_t> fit, big;
    std::vector<uint8_t> is_big(B, 0);
    size_t so = 0, dy = 0;
    p->res_fit_n = p->res_fit_m = p->res_fit_e = 0;
    for (size_t b = 0; b < B; ++b) {
        const int n = v0[b + 1] - v0[b], m = f0[b + 1] - f0[b], e = e0[b + 1] - e0[b];
        off[b] = (int64_t)so; off[B + b] = (int64_t)dy;
        if (!instance_fits_lds(n, m, e)) { big.push_back((int32_t)b); is_big[b] = 1; continue; }
        fit.push_back((int32_t)b);
        if (n > p->res_fit_n) p->res_fit_n = n;
        if (m > p->res_fit_m) p->res_fit_m = m;
        if (e > p->res_fit_e) p->res_fit_e = e;
        const BlobLayout bl = blob_layout(n, m, e);
        so += bl.stat_bytes; dy += bl.dyn_bytes;
    }
    p->res_nfit = (int)fit.size(); p->res_nbig = (int)big.size();
    { int st_ = pdp_dev_alloc((void **)&p->res_fit_list, (fit.size() + 1) * 4); if (st_ != PDP_OK) return st_; }
    { int st_ = pdp_dev_alloc((void **)&p->res_big_list, (big.size() + 1) * 4); if (st_ != PDP_OK) return st_; }
    { int st_ = pdp_dev_alloc((void **)&p->res_is_big, B + 1); if (st_ != PDP_OK) return st_; }
    if (!fit.empty()) PDP_HIP_CHECK(hipMemcpy(p->res_fit_list, fit.data(), fit.size() * 4, hipMemcpyHostToDevice));
    if (!big.empty()) PDP_HIP_CHECK(hipMemcpy(p->res_big_list, big.data(), big.size() * 4, hipMemcpyHostToDevice));
    PDP_HIP_CHECK(hipMemcpy(p->res_is_big, is_big.data(), B, hipMemcpyHostToDevice));
    p->res_stat_bytes = so + 16; p->res_dyn_bytes = dy + 16;
    { int st_ = pdp_dev_alloc((void **)&p->res_stat, p->res_stat_bytes); if (st_ != PDP_OK) return st_; }
    { int st_ = pdp_dev_alloc((void **)&p->res_dyn[0], p->res_dyn_bytes); if (st_ != PDP_OK) return st_; }
    { int st_ = pdp_dev_alloc((void **)&p->res_dyn[1], p->res_dyn_bytes); if (st_ != PDP_OK) return st_; }
    { int st_ = pdp_dev_alloc((void **)&p->res_prev_slots, (E + 4) * sizeof(float)); if (st_ != PDP_OK) return st_; }
    p->res_static_built = 0;
    { int st_ = pdp_dev_alloc((void **)&p->res_stat_off, 2 * B * sizeof(int64_t)); if (st_ != PDP_OK) return st_; }   // (set last: marks the preparation complete)
    PDP_HIP_CHECK(hipMemcpy(p->res_stat_off, off.data(), 2 * B * sizeof(int64_t), hipMemcpyHostToDevice));
    return PDP_OK;
}

// workspaces of the HBM-resident kernel (shared by the host-driven loop, the exact single-instance launch and the lock-step launch)
static int hbm_workspaces(pdp_problem *p, SolveParams &sp, bool records = true)
{
    for (int i = 0; i < 4; ++i) sp.ws_e[i] = p->ws_e[i];
    sp.ws_f = p->ws_f[0];
    for (int i = 0; i < 6; ++i) sp.ws_v[i] = p->ws_v[i];
    if (!p->solve_extra_v) { int st_ = pdp_dev_alloc((void **)&p->solve_extra_v, sizeof(float) * (size_t)p->V); if (st_ != PDP_OK) return st_; }
    sp.ws_v[6] = p->solve_extra_v;
    for (int i = 0; i < 3; ++i) sp.ws_vi[i] = p->ws_vi[i];
    sp.ws_fu[0] = p->ws_fu[0]; sp.ws_fu[1] = p->ws_fu[1];
    if (records) {
        if (!p->solve_rec) { int st_ = pdp_dev_alloc((void **)&p->solve_rec, sizeof(float) * 4 * ((size_t)p->E + 4)); if (st_ != PDP_OK) return st_; }
        sp.ws_r = p->solve_rec;
    }
    return PDP_OK;
}

// LDS-resident path: every launch of the call is enqueued up front (import, then per chunk: pass 1, poison decision, replay
// list, replay, speculation check); the host reads one control block at the end.
static bool lockstep_possible(const pdp_problem *p, const pdp_solve_args *a);
static int sp_solve_resident(pdp_problem *p, pdp_solve_args *a, hipStream_t st, bool force, size_t lds, int nt_lds, int C)
{
    const bool rf = a->model == PDP_MODEL_REINFORCE;
    const int T = a->iterations;
    const size_t E = p->E, V = p->V, F = p->F, B = p->B;
    // The chunk schedule: lengths C0, C, C, ... (C0 = PDP_SOLVE_FIRST_CHUNK, default 2 C).  What a short chunk bounds is the poison replay, and a
    // NaN needs a decimation, which needs a converged instance: the first sweeps of a call are the least likely to hold one, and a launch of 50
    // sweeps costs 4.07 ms where two of 25 cost 4.23 (the chip drains and refills once less).  With isolated instances there is no poison and
    // nothing to replay: the whole loop is one launch (100 sweeps: 8.78 ms against 4 x 2.30).
    std::vector<int> cstart, clen;
    {
        int c0 = a->isolate_instances ? T : 2 * C;
        if (const char *env = getenv("PDP_SOLVE_FIRST_CHUNK")) { const int v = atoi(env); if (v > 0) c0 = v; }
        for (int d = 0; d < T;) { const int c = std::min(T - d, d == 0 ? c0 : C); cstart.push_back(d); clen.push_back(c); d += c; }
    }
    const int nchunks = (int)clen.size();
    const int64_t *stat_off = p->res_stat_off, *dyn_off = p->res_stat_off + B;
    // ---- control blocks, speculation record, replay list; call-entry snapshot for the failure path ---------------------------
    const size_t ctl_bytes = (size_t)nchunks * sizeof(SolveCtl) + sizeof(SolveCall) + 2 * (size_t)T * 4 + 2 * B * 4 + 64 + 2 * (B + 16) * 4 +
                             ((rf || force) ? 2 * (E + 4) * sizeof(float) : 0) +   // slot-major force column (Reinforce: two, it changes from chunk to chunk)
                             ((B + 63) & ~(size_t)63);                          // ghost flags
    int status = ensure_bytes(&p->res_ctl, &p->res_ctl_bytes, ctl_bytes);
    if (status != PDP_OK) return status;
    SolveCtl *ctl = (SolveCtl *)p->res_ctl;
    SolveCall *call = (SolveCall *)(ctl + nchunks);
    uint32_t *spec = (uint32_t *)(call + 1);                 // [T] used | [T] zero
    int32_t *last_event = (int32_t *)(spec + 2 * (size_t)T);
    int32_t *replay_list = last_event + B;
    const size_t snap_floats = 3 * E + 2 * E + V + F + V + B + E + E + B;
    const size_t snap_bytes = snap_floats * 4 + ((B + 63) & ~(size_t)63);
    status = ensure_bytes(&p->solve_blob, &p->solve_blob_bytes, snap_bytes + 64);
    if (status != PDP_OK) return status;
    const size_t host_words = ((size_t)nchunks * sizeof(SolveCtl) + sizeof(SolveCall)) / 4;
    if (p->solve_host_words < host_words) {
        if (p->solve_host) (void)hipHostFree(p->solve_host);
        p->solve_host = nullptr; p->solve_host_words = 0;
        PDP_HIP_CHECK(hipHostMalloc((void **)&p->solve_host, host_words * 4));
        p->solve_host_words = host_words;
    }
    SolveSnapshot snap0;
    {
        float *f = (float *)p->solve_blob;
        snap0.q = f; f += 3 * E; snap0.fs = f; f += 2 * E; snap0.av = f; f += V; snap0.af = f; f += F; snap0.sol = f; f += V;
        snap0.sat = f; f += B; snap0.emask = f; f += E; snap0.prev = f; f += E; snap0.cnt = f; f += B; snap0.amask = (uint8_t *)f;
    }
    const int had_prev0 = a->decimator->has_prev, had_emask0 = p->has_edge_mask;
    // (with big instances in the batch the HBM-resident kernel works in place on the caller's arrays: everything is kept then)
    // (and a small batch whose speculation fails reruns in the lock-step launch from the restored q / fs: they are kept for it)
    const int snap_skip = p->res_nbig ? 0 : (((a->inputs_disposable && !lockstep_possible(p, a)) ? 1 : 0) | (had_prev0 ? 0 : 2) | (had_emask0 ? 0 : 4));
    // (all that is left of it are the arrays k_solve_import reads anyway, and every instance goes through the import: it takes the snapshot)
    const bool snap_in_import = snap_skip == 7 && getenv("PDP_SOLVE_SNAPSHOT_COPY") == nullptr;
    if (!snap_in_import) { status = snapshot_copy(p, a, snap0, true, st, snap_skip); if (status != PDP_OK) return status; }

    if (rf) {
        PDP_HIP_CHECK(hipFuncSetAttribute((const void *)k_sp_solve_lds<true, false, true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        PDP_HIP_CHECK(hipFuncSetAttribute((const void *)k_sp_solve_lds<true, false, true, true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        PDP_HIP_CHECK(hipFuncSetAttribute((const void *)k_sp_solve_lds<true, true, true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    } else if (force) {
        PDP_HIP_CHECK(hipFuncSetAttribute((const void *)k_sp_solve_lds<true, false>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        PDP_HIP_CHECK(hipFuncSetAttribute((const void *)k_sp_solve_lds<true, false, false, true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        PDP_HIP_CHECK(hipFuncSetAttribute((const void *)k_sp_solve_lds<true, true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    } else {
        PDP_HIP_CHECK(hipFuncSetAttribute((const void *)k_sp_solve_lds<false, false>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        PDP_HIP_CHECK(hipFuncSetAttribute((const void *)k_sp_solve_lds<false, false, false, true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        PDP_HIP_CHECK(hipFuncSetAttribute((const void *)k_sp_solve_lds<false, true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    }
    a->used_lds_host = 1;
    // optional: HIP events around every solver launch (the caller asks for kernel times, e.g. the benchmark's roofline line)
    const bool timed = a->time_kernels != 0;
    // An event record between two dependent launches costs 3-6 us of stream time (tools/micro/event_gap.hip), so there are as few as the two
    // figures need: [2k] / [2k + 1] around chunk k's launch, [2 nchunks] behind the last chunk's control kernels (a replay is timed from
    // the end of its chunk's launch to the next chunk's start); no system-scope fence on them (nothing on the host reads what the kernels wrote
    // until the stream is drained).
    const int n_events = 2 * nchunks + 1;
    if (timed && p->res_events_n < n_events) {
        hipEvent_t *ev = (hipEvent_t *)realloc(p->res_events, sizeof(hipEvent_t) * (size_t)n_events);
        PDP_REQUIRE(ev, "out of host memory");
        p->res_events = ev;
        for (; p->res_events_n < n_events; ++p->res_events_n)
            PDP_HIP_CHECK(hipEventCreateWithFlags(&p->res_events[p->res_events_n], hipEventDisableSystemFence));
    }

    const int nfit = p->res_nfit, nbig = p->res_nbig;
    const int32_t *fit_list = nbig ? p->res_fit_list : nullptr;                 // all instances fit: instance = block index
    // where the rest of the control block lies (host arithmetic only): the force columns (Reinforce / an external force), the ghost flags, the
    // dispatch order of pass 1 -- and whether pass 1 is risk-ordered at all
    float *frc_buf[2] = {nullptr, nullptr};
    if (rf || force) { frc_buf[0] = (float *)(((uintptr_t)(replay_list + B) + 15) & ~(uintptr_t)15); frc_buf[1] = frc_buf[0] + E + 4; }
    uint8_t *ghost_flag = (rf || force) ? (uint8_t *)(frc_buf[1] + E + 4) : (uint8_t *)(((uintptr_t)(replay_list + B) + 15) & ~(uintptr_t)15);
    uint32_t *risk = (uint32_t *)(((uintptr_t)(ghost_flag + ((B + 63) & ~(size_t)63)) + 15) & ~(uintptr_t)15);
    int32_t *order = (int32_t *)(risk + B + 8);
    // ticketed LDS-resident pass (see k_sp_solve_lds): on for mixed batches; PDP_SOLVE_TICKETS=<percent of over-provisioning>, 0 = off
    int ticket_extra = nbig > 0 ? 25 : 0;
    if (const char *env = getenv("PDP_SOLVE_TICKETS")) ticket_extra = atoi(env);
    // a batch that is LDS-resident as a whole: pass 1 takes its instances in the order k_solve_finish leaves (PDP_SOLVE_NO_RISK_ORDER=1: block index)
    const bool risk_order = nbig == 0 && ticket_extra <= 0 && !fit_list && nchunks > 1 && !a->isolate_instances && getenv("PDP_SOLVE_NO_RISK_ORDER") == nullptr;
    {
        const int words = 2 * T, cover = (int)B > words ? (int)B : words;
        hipLaunchKernelGGL(k_solve_ctl_init, dim3(((cover > nchunks ? cover : nchunks) + 255) / 256), dim3(256), 0, st, ctl, nchunks, call, spec, words, ghost_flag, (int)B,
                           risk_order ? order : (int32_t *)nullptr, risk);
    }
    // staging area of the import: five columns of the largest fitting instance
    int stage_cap = (p->res_fit_e + 7) & ~7;
    const size_t stage_per_edge = a->decimator->has_prev ? 14 : 10;                // q_u, survey, clause word (16 bits) [, previous survey]
    if ((size_t)stage_cap * stage_per_edge > 159 * 1024) stage_cap = 0;            // (0: gather from global memory)
    if (getenv("PDP_SOLVE_IMPORT_GATHER")) stage_cap = 0;
    const size_t stage_bytes = (size_t)stage_cap * stage_per_edge;
    if (stage_bytes > 64 * 1024) PDP_HIP_CHECK(hipFuncSetAttribute((const void *)k_solve_import, hipFuncAttributeMaxDynamicSharedMemorySize, (int)stage_bytes));
    hipLaunchKernelGGL(k_solve_import, dim3(nfit), dim3(256), stage_bytes, st, make_view(p), (const float *)a->q, (const float *)a->fs, (const uint8_t *)a->active_mask,
                       (const float *)a->decimator->prev, (const float *)a->decimator->counters, a->decimator->has_prev, p->res_static_built ? 0 : 1,
                       p->res_stat, p->res_dyn[0], stat_off, dyn_off, p->res_prev_slots, fit_list, stage_cap, (force || rf) ? (SolveCall *)nullptr : call,
                       p->has_edge_mask ? 1 : 0, snap_in_import ? snap0 : SolveSnapshot{});
    PDP_LAUNCH_CHECK();
    p->res_static_built = 1;

    SolveParams sp;
    memset(&sp, 0, sizeof(sp));
    sp.tol = a->tolerance; sp.t_max = a->t_max; sp.pi = a->pi;
    sp.q = a->q; sp.fs = a->fs; sp.amask = a->active_mask; sp.src_fs = a->fs;
    sp.prev = a->decimator->prev; sp.counters = a->decimator->counters;
    sp.check_termination = a->check_termination;
    sp.coins = a->coins; sp.dprob = a->decimation_probability;
    sp.rf = rf ? 1 : 0; sp.isolate = a->isolate_instances ? 1 : 0;
    sp.adopt_poison = getenv("PDP_SOLVE_NO_ADOPT") ? 0 : 1;
    sp.debug_ghost_inject = getenv("PDP_DEBUG_GHOST_INJECT") ? 1 : 0;
    sp.no_scorer_reuse = getenv("PDP_SOLVE_NO_SCORER_REUSE") ? 1 : 0;
    sp.no_event_look = getenv("PDP_SOLVE_NO_EVENT_LOOK") ? 1 : 0;
    sp.rf_no_fused_step = getenv("PDP_SOLVE_RF_NO_FUSED_STEP") ? 1 : 0;
    if (rf || force) {
        // the external force as a 2-bit code in the slot word (0, +1, -1, NaN; any other value raises the violation flag: the call fails over
        // to the step-wise loop, which takes the column as it is).  The SP triple only reads it (one column for all chunks).
        hipLaunchKernelGGL(k_force_import, dim3(p->B), dim3(256), 0, st, make_view(p), (const float *)a->fs, frc_buf[0], ctl, (const uint8_t *)a->active_mask);
    }
    sp.ghost_flag = ghost_flag;                             // (cleared by k_solve_ctl_init, like the identity dispatch order)
    sp.last_event = last_event; sp.inst_list = replay_list;
    sp.call = call; sp.stat = p->res_stat; sp.stat_off = stat_off; sp.dyn_off = dyn_off;
    sp.fit_list = fit_list;
    // instances that do not fit the LDS: the HBM-resident kernel on their list, under the same device-side control; it works in place, so
    // the state they enter a chunk with is saved for the NaN-poison replay
    BigSnap big_live, big_snap;
    if (nbig) {
        { const int st_ = hbm_workspaces(p, sp); if (st_ != PDP_OK) return st_; }
        sp.big_list = p->res_big_list; sp.hbm_device_ctl = 1;
        status = ensure_bytes(&p->res_big_snap, &p->res_big_snap_bytes, snap_bytes + 64);
        if (status != PDP_OK) return status;
        float *f = (float *)p->res_big_snap;
        big_snap.q = f; f += 3 * E; big_snap.fs = f; f += 2 * E; big_snap.av = f; f += V; big_snap.af = f; f += F; big_snap.sol = f; f += V;
        big_snap.sat = f; f += B; big_snap.emask = f; f += E; big_snap.prev = f; f += E; big_snap.cnt = f; f += B; big_snap.amask = (uint8_t *)f;
        big_live.q = a->q; big_live.fs = a->fs; big_live.av = p->av; big_live.af = p->af; big_live.sol = p->sol; big_live.sat = p->is_sat;
        big_live.emask = p->emask; big_live.prev = a->decimator->prev; big_live.cnt = a->decimator->counters; big_live.amask = a->active_mask;
        if (!p->res_side_stream) {
            int prio_lo = 0, prio_hi = 0;
            PDP_HIP_CHECK(hipDeviceGetStreamPriorityRange(&prio_lo, &prio_hi));
            PDP_HIP_CHECK(hipStreamCreateWithPriority(&p->res_side_stream, hipStreamNonBlocking, prio_hi));
            for (int i = 0; i < 2; ++i) PDP_HIP_CHECK(hipEventCreateWithFlags(&p->res_side_ev[i], hipEventDisableTiming));
        }
    }
    hipStream_t side = p->res_side_stream;
    // an error return between here and the final synchronisation must not leave work on the side stream that still reads the caller's arrays
    struct SideJoin { hipStream_t s; bool on; ~SideJoin() { if (on && s) (void)hipStreamSynchronize(s); } } side_join{side, nbig > 0};
    auto launch_big = [&](const SolveParams &spx, hipStream_t s_) { return launch_hbm(p, spx, nbig, s_); };
    if (ticket_extra > 0 && !sp.fit_list) sp.fit_list = p->res_fit_list;        // (tickets index the list; with no big instance it holds every instance)
    if (risk_order) { sp.risk = risk; sp.fit_list = order; }
    const bool listed = sp.fit_list != nullptr;                                 // pass 1 runs the LISTED instantiation
    const int big_copy_wgs = nbig >= 256 ? 1 : (256 / (nbig > 0 ? nbig : 1) < 32 ? 256 / (nbig > 0 ? nbig : 1) : 32);      // workgroups per instance of the save / restore copies
    if (const char *env = getenv("PDP_DEBUG_SKIP")) sp.debug_skip = atoi(env);
    int done = 0;
    const bool xch = p->exchange != nullptr;              // (pdp_sp_solve admits the hook only for batches this loop takes whole)
    bool x_poisoned = false, x_replay = false, x_stop = false;        // host mirrors of call->poisoned_all / ctl->do_replay / call->stop (all derived from merged words)
    for (int k = 0; k < nchunks; ++k) {
        const int c = clen[k];
        x_replay = false;
        sp.T = c; sp.chunk_start = done; sp.final_chunk = (done + c >= T) ? 1 : 0;
        sp.has_prev = (k == 0) ? a->decimator->has_prev : 1;
        sp.has_edge_mask = (k == 0) ? p->has_edge_mask : 1;
        sp.prev_slots = (k == 0 && a->decimator->has_prev) ? p->res_prev_slots : nullptr;
        sp.ctl = ctl + k; sp.spec_used = spec + done; sp.spec_zero = spec + T + done;
        sp.dyn_in = p->res_dyn[k & 1]; sp.dyn_out = p->res_dyn[(k + 1) & 1];
        sp.frc_in = rf ? frc_buf[k & 1] : frc_buf[0]; sp.frc_out = rf ? frc_buf[(k + 1) & 1] : frc_buf[1];
        if (nbig) {
            // pass 1 of the big instances goes FIRST, on the side stream: its few workgroups take their CUs before the LDS-resident kernel
            // fills the chip, and the two kernels overlap (they meet in front of k_solve_post)
            sp.nan_iter = &ctl[k].nan_iter; sp.w_perm_zero = &ctl[k].perm_zero; sp.w_iters_run = &ctl[k].iters_run; sp.w_violation = &ctl[k].violation;
            sp.hbm_replay = 0;
            PDP_HIP_CHECK(hipEventRecord(p->res_side_ev[0], st)); PDP_HIP_CHECK(hipStreamWaitEvent(side, p->res_side_ev[0], 0));
            hipLaunchKernelGGL(k_big_state, dim3(nbig, big_copy_wgs), dim3(256), 0, side, make_view(p), (const int32_t *)p->res_big_list, big_live, big_snap, 0,
                               (const SolveCtl *)(ctl + k), (const SolveCall *)call);
            { const int st_ = launch_big(sp, side); if (st_ != PDP_OK) return st_; }
        }
        for (int pass = 0; pass < 2; ++pass) {
            if (nbig && pass == 1) {
                // replay of the big instances: restored and rerun on the side stream while the selective replay of the small ones runs
                sp.hbm_replay = 1;
                PDP_HIP_CHECK(hipEventRecord(p->res_side_ev[0], st)); PDP_HIP_CHECK(hipStreamWaitEvent(side, p->res_side_ev[0], 0));
                hipLaunchKernelGGL(k_big_state, dim3(nbig, big_copy_wgs), dim3(256), 0, side, make_view(p), (const int32_t *)p->res_big_list, big_live, big_snap, 1,
                                   (const SolveCtl *)(ctl + k), (const SolveCall *)call);
                { const int st_ = launch_big(sp, side); if (st_ != PDP_OK) return st_; }
            }
            if (timed && pass == 0) PDP_HIP_CHECK(hipEventRecord(p->res_events[2 * k], st));
            sp.lds_tickets = (pass == 0 && ticket_extra > 0) ? nfit : 0;
            const int grid_lds = sp.lds_tickets ? nfit + (int)(((int64_t)nfit * ticket_extra + 99) / 100) : nfit;
            if (pass == 0)
                pdp_note_kernel(PDP_KN_SP_SOLVE, listed ? (rf ? "k_sp_solve_lds<true, false, true, true>" : force ? "k_sp_solve_lds<true, false, false, true>" : "k_sp_solve_lds<false, false, false, true>")
                                                        : (rf ? "k_sp_solve_lds<true, false, true, false>" : force ? "k_sp_solve_lds<true, false, false, false>" : "k_sp_solve_lds<false, false, false, false>"));
            else
                pdp_note_kernel(PDP_KN_SP_REPLAY, rf ? "k_sp_solve_lds<true, true, true, false>" : force ? "k_sp_solve_lds<true, true, false, false>" : "k_sp_solve_lds<false, true, false, false>");
            if (pass == 0 && listed) {
                if (rf) hipLaunchKernelGGL((k_sp_solve_lds<true, false, true, true>), dim3(grid_lds), dim3(nt_lds), lds, st, make_view(p), sp);
                else if (force) hipLaunchKernelGGL((k_sp_solve_lds<true, false, false, true>), dim3(grid_lds), dim3(nt_lds), lds, st, make_view(p), sp);
                else hipLaunchKernelGGL((k_sp_solve_lds<false, false, false, true>), dim3(grid_lds), dim3(nt_lds), lds, st, make_view(p), sp);
            } else
            if (rf && pass == 0) hipLaunchKernelGGL((k_sp_solve_lds<true, false, true>), dim3(grid_lds), dim3(nt_lds), lds, st, make_view(p), sp);
            else if (rf) hipLaunchKernelGGL((k_sp_solve_lds<true, true, true>), dim3(nfit), dim3(nt_lds), lds, st, make_view(p), sp);
            else if (force && pass == 0) hipLaunchKernelGGL((k_sp_solve_lds<true, false>), dim3(grid_lds), dim3(nt_lds), lds, st, make_view(p), sp);
            else if (force) hipLaunchKernelGGL((k_sp_solve_lds<true, true>), dim3(nfit), dim3(nt_lds), lds, st, make_view(p), sp);
            else if (pass == 0) hipLaunchKernelGGL((k_sp_solve_lds<false, false>), dim3(grid_lds), dim3(nt_lds), lds, st, make_view(p), sp);
            else hipLaunchKernelGGL((k_sp_solve_lds<false, true>), dim3(nfit), dim3(nt_lds), lds, st, make_view(p), sp);
            if (timed && pass == 0) PDP_HIP_CHECK(hipEventRecord(p->res_events[2 * k + 1], st));
            if (nbig && pass == 1) { PDP_HIP_CHECK(hipEventRecord(p->res_side_ev[1], side)); PDP_HIP_CHECK(hipStreamWaitEvent(st, p->res_side_ev[1], 0)); }
            if (nbig && pass == 0) { PDP_HIP_CHECK(hipEventRecord(p->res_side_ev[1], side)); PDP_HIP_CHECK(hipStreamWaitEvent(st, p->res_side_ev[1], 0)); }   // join: pass 1 of the big instances
            if (xch && !x_stop && (pass == 0 || x_replay)) {
                // A coupled forward spread over several processes: what the reference reduces over the WHOLE batch -- the first NaN sweep, the
                // exact-zero record of the batch-global minimum, the executed sweeps -- is completed across the parts before the device-side
                // control of this chunk reads it (pass 0), and again after a poison replay (the replayed instances run on).
                SolveCtl hc;
                PDP_HIP_CHECK(hipMemcpyAsync(&hc, ctl + k, sizeof(hc), hipMemcpyDeviceToHost, st));
                std::vector<uint32_t> bits(2 * (size_t)c);
                if (pass == 0) {
                    PDP_HIP_CHECK(hipMemcpyAsync(bits.data(), sp.spec_used, (size_t)c * 4, hipMemcpyDeviceToHost, st));
                    PDP_HIP_CHECK(hipMemcpyAsync(bits.data() + c, sp.spec_zero, (size_t)c * 4, hipMemcpyDeviceToHost, st));
                }
                PDP_HIP_CHECK(hipStreamSynchronize(st));
                if (pass == 0 || x_replay) {
                    const uint32_t mins[2] = {hc.nan_iter, hc.perm_zero}, maxs[1] = {hc.iters_run};
                    std::vector<uint32_t> ors(1 + (pass == 0 ? 2 * (size_t)c : 0));
                    ors[0] = hc.violation;
                    if (pass == 0) for (int t = 0; t < 2 * c; ++t) ors[1 + t] = bits[t];
                    uint32_t *m = nullptr;
                    { const int st_ = pdp_exchange_call(p, mins, pass == 0 ? 2 : 0, maxs, 1, ors.data(), (int)ors.size(), &m); if (st_ != PDP_OK) return st_; }
                    const uint32_t *mm = m, *mx = m + (pass == 0 ? 2 : 0), *mo = mx + 1;
                    if (pass == 0) { hc.nan_iter = mm[0]; hc.perm_zero = mm[1]; }
                    hc.iters_run = mx[0]; hc.violation = mo[0];
                    // (only the four merged words go back: the kernels own the rest of the block)
                    PDP_HIP_CHECK(hipMemcpyAsync(&ctl[k].nan_iter, &hc.nan_iter, 4 * sizeof(uint32_t), hipMemcpyHostToDevice, st));
                    if (pass == 0) {
                        PDP_HIP_CHECK(hipMemcpyAsync((void *)sp.spec_used, mo + 1, (size_t)c * 4, hipMemcpyHostToDevice, st));
                        PDP_HIP_CHECK(hipMemcpyAsync((void *)sp.spec_zero, mo + 1 + c, (size_t)c * 4, hipMemcpyHostToDevice, st));
                    }
                    PDP_HIP_CHECK(hipStreamSynchronize(st));          // (the staging block is reused by the next exchange)
                    if (pass == 0) { x_replay = !x_poisoned && hc.nan_iter < (uint32_t)c; if (x_replay) x_poisoned = true; }
                    if (pass == 1 || !x_replay) { if (hc.iters_run < (uint32_t)c) x_stop = true; }
                }
            }
            if (pass == 0) {
                hipLaunchKernelGGL(k_solve_post, dim3(1), dim3(1024), 0, st, ctl + k, call, c, (int)a->isolate_instances, p->B, (const int32_t *)last_event, replay_list,
                                   (const uint8_t *)(nbig ? p->res_is_big : nullptr));
            }
        }
        hipLaunchKernelGGL(k_solve_finish, dim3(1), dim3(1024), 0, st, ctl + k, call, (const uint32_t *)sp.spec_used, (const uint32_t *)sp.spec_zero, c, done, (int)a->isolate_instances,
                           p->B, (const uint32_t *)risk, (risk_order && k + 1 < nchunks) ? order : (int32_t *)nullptr,
                           (k + 1 == nchunks && !a->isolate_instances) ? (const uint8_t *)ghost_flag : (const uint8_t *)nullptr);
        PDP_LAUNCH_CHECK();
        done += c;
    }
    if (timed) PDP_HIP_CHECK(hipEventRecord(p->res_events[2 * nchunks], st));
    PDP_HIP_CHECK(hipMemcpyAsync(p->solve_host, ctl, host_words * 4, hipMemcpyDeviceToHost, st));
    PDP_HIP_CHECK(hipStreamSynchronize(st));
    side_join.on = false;                                  // (every side-stream launch was joined into the main stream, which is drained)
    const SolveCtl *hctl = (const SolveCtl *)p->solve_host;
    SolveCall *hcall = (SolveCall *)(hctl + nchunks);
    if (xch) {
        // the parts agree on the outcome: a failed speculation anywhere (the ghost check is local) fails the forward everywhere
        const uint32_t maxs[1] = {hcall->total_iters}, ors[2] = {hcall->fail, hcall->force_seen};
        uint32_t *m = nullptr;
        { const int st_ = pdp_exchange_call(p, nullptr, 0, maxs, 1, ors, 2, &m); if (st_ != PDP_OK) return st_; }
        hcall->total_iters = m[0]; hcall->fail = m[1]; hcall->force_seen = m[2];
    }
    const bool debug = getenv("PDP_DEBUG") != nullptr;
    int launches = 0, replays = 0;
    float solve_ms = 0.0f, replay_ms = 0.0f;
    for (int k = 0; k < nchunks; ++k) {
        if (hctl[k].iters_run == 0 && cstart[k] >= (int)hcall->total_iters && k > 0) break;     // launches after the global early exit return at once
        const bool replayed = hctl[k].do_replay && hctl[k].replay_count;
        launches += 1; replays += replayed ? 1 : 0;
        if (timed) {
            float ms = 0.0f;
            PDP_HIP_CHECK(hipEventElapsedTime(&ms, p->res_events[2 * k], p->res_events[2 * k + 1])); solve_ms += ms;
            if (replayed) { PDP_HIP_CHECK(hipEventElapsedTime(&ms, p->res_events[2 * k + 1], p->res_events[2 * k + 2])); replay_ms += ms; }
        }
        if (debug)
            fprintf(stderr, "[pdp_sp_solve] chunk@%d violation=%u perm_from=%u nan_iter=%u poison_from=%d replayed=%u iters=%u lds=%zu\n", cstart[k],
                    hctl[k].violation, hctl[k].perm_zero, hctl[k].nan_iter, hctl[k].poison_from, hctl[k].do_replay ? hctl[k].replay_count : 0u, hctl[k].iters_run, lds);
    }
    a->kernel_launches_host = launches; a->replay_launches_host = replays;
    a->solve_kernel_ms_host = solve_ms; a->replay_kernel_ms_host = replay_ms;
    if (!force && !rf && hcall->force_seen) {
        // the force-free instantiation met a force column: its launches returned at once; what the big instances' kernel did in place is undone
        status = snapshot_copy(p, a, snap0, false, st, snap_skip);
        if (status == PDP_OK) status = hipStreamSynchronize(st) == hipSuccess ? PDP_OK : PDP_ERR_HIP;
        a->decimator->has_prev = had_prev0; p->has_edge_mask = had_emask0;
        return status == PDP_OK ? PDP_RETRY_WITH_FORCE : status;
    }
    if (hcall->fail) {
        // leave the caller's state exactly as it was at call entry so that it can rerun the batch step-wise
        status = snapshot_copy(p, a, snap0, false, st, snap_skip);
        if (status == PDP_OK) status = hipStreamSynchronize(st) == hipSuccess ? PDP_OK : PDP_ERR_HIP;
        a->decimator->has_prev = had_prev0; p->has_edge_mask = had_emask0;
        if (status != PDP_OK) return status;
        pdp_set_error("persistent solve: a cross-instance coupling of the reference became active (batch-global min != 0); "
                      "state restored, rerun the batch step-wise");
        return PDP_ERR_SPECULATION;
    }
    a->decimator->has_prev = 1;
    p->has_edge_mask = 1;
    a->iterations_run_host = (int)hcall->total_iters;
    return PDP_OK;
}

// Lock-step solve of a small batch -- up to 1 024 instances -- (k_sp_solve<256, false, true>): exact reference semantics without speculation -- what the batch takes when
// the speculation failed (no instance supplied the exact zero: batches of a few instances), and what replicated batches with non-identical
// replicas take (random initial state: the replicas couple through the termination rule).  One launch, no snapshot.
static bool lockstep_possible(const pdp_problem *p, const pdp_solve_args *a)
{
    if (a->isolate_instances || getenv("PDP_SOLVE_NO_LOCKSTEP")) return false;
    if (a->model == PDP_MODEL_REINFORCE && p->R > 1) return false;      // (Reinforce with batch replication: step-wise)
    // every workgroup must be resident at once (they wait for each other): what the device holds of this kernel, and the mailbox capacity
    static int per_cu_dev[64] = {0};                              // per device id, like pdp_device_cus
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return false;
    if (!per_cu_dev[dev]) { int v = 0; per_cu_dev[dev] = (hipOccupancyMaxActiveBlocksPerMultiprocessor(&v, (const void *)k_sp_solve<256, false, true>, 256, 0) == hipSuccess && v > 0) ? v : 1; }
    const int per_cu = per_cu_dev[dev];
    const long resident = (long)pdp_device_cus() * (per_cu < 4 ? per_cu : 4);
    return p->B <= resident && p->B <= PDP_LOCK_MAX && p->max_e <= 65536;     // (an instance is one workgroup's work)
}
static int sp_solve_lockstep(pdp_problem *p, pdp_solve_args *a, hipStream_t st)
{
    // call-entry snapshot: the one thing the launch cannot do itself is the NaN blend of an inactive instance (ghost_bad in k_sp_solve)
    const size_t E = p->E, V = p->V, F = p->F, B = p->B;
    const size_t snap_floats = 3 * E + 2 * E + V + F + V + B + E + E + B;
    const size_t snap_bytes = snap_floats * 4 + ((B + 63) & ~(size_t)63);
    { const int st_ = ensure_bytes(&p->solve_blob, &p->solve_blob_bytes, snap_bytes + 64); if (st_ != PDP_OK) return st_; }
    SolveSnapshot snap0;
    {
        float *f = (float *)p->solve_blob;
        snap0.q = f; f += 3 * E; snap0.fs = f; f += 2 * E; snap0.av = f; f += V; snap0.af = f; f += F; snap0.sol = f; f += V;
        snap0.sat = f; f += B; snap0.emask = f; f += E; snap0.prev = f; f += E; snap0.cnt = f; f += B; snap0.amask = (uint8_t *)f;
    }
    const int had_prev0 = a->decimator->has_prev, had_emask0 = p->has_edge_mask;
    { const int st_ = snapshot_copy(p, a, snap0, true, st); if (st_ != PDP_OK) return st_; }
    SolveParams sp;
    memset(&sp, 0, sizeof(sp));
    sp.tol = a->tolerance; sp.t_max = a->t_max; sp.pi = a->pi;
    sp.q = a->q; sp.fs = a->fs; sp.amask = a->active_mask;
    sp.prev = a->decimator->prev; sp.counters = a->decimator->counters;
    sp.check_termination = a->check_termination;
    sp.w_perm_zero = p->flags + FL_PERM_ZERO; sp.w_iters_run = p->flags + FL_ITERS_RUN; sp.w_violation = p->flags + FL_SPEC_VIOLATION;
    sp.nan_iter = p->flags + FL_N_SEL; sp.spec_used = nullptr; sp.spec_zero = nullptr;
    sp.T = a->iterations; sp.has_prev = a->decimator->has_prev; sp.has_edge_mask = p->has_edge_mask; sp.final_chunk = 1; sp.poison_from = 0x7fffffff;
    sp.rf = a->model == PDP_MODEL_REINFORCE ? 1 : 0; sp.coins = a->coins; sp.dprob = a->decimation_probability; sp.chunk_start = 0;
    { const int st_ = hbm_workspaces(p, sp); if (st_ != PDP_OK) return st_; }
    if (!p->team_ws) { int st_ = pdp_dev_alloc((void **)&p->team_ws, sizeof(uint32_t) * 256 * PDP_TEAM_WORDS); if (st_ != PDP_OK) return st_; }
    PDP_HIP_CHECK(hipMemsetAsync(p->team_ws, 0, sizeof(uint32_t) * PDP_TEAM_WORDS, st));
    sp.team_ws = p->team_ws; sp.spin_limit = pdp_spin_limit();
    // (test hook: PDP_DEBUG_LOCK_EXTRA=1 makes the batch barrier wait for a workgroup that does not exist -- the bounded wait must give up,
    //  report a violation, and the call must fail over instead of hanging)
    sp.lock_size = getenv("PDP_DEBUG_LOCK_EXTRA") ? p->B + 1 : 0;
    PDP_HIP_CHECK(hipMemsetAsync(p->flags + FL_PERM_ZERO, 0xff, sizeof(uint32_t), st));
    PDP_HIP_CHECK(hipMemsetAsync(p->flags + FL_SPEC_VIOLATION, 0, sizeof(uint32_t) * 2, st));
    hipLaunchKernelGGL((k_sp_solve<256, false, true>), dim3(p->B), dim3(256), 0, st, make_view(p), sp);
    PDP_LAUNCH_CHECK();
    PDP_HIP_CHECK(hipMemcpyAsync(p->flags_host, p->flags, FL_COUNT * sizeof(uint32_t), hipMemcpyDeviceToHost, st));
    PDP_HIP_CHECK(hipStreamSynchronize(st));
    if (p->flags_host[FL_SPEC_VIOLATION] != 0u) {
        int status = snapshot_copy(p, a, snap0, false, st);
        if (status == PDP_OK) status = hipStreamSynchronize(st) == hipSuccess ? PDP_OK : PDP_ERR_HIP;
        if (status != PDP_OK) return status;
        a->decimator->has_prev = had_prev0; p->has_edge_mask = had_emask0;
        pdp_set_error("persistent solve: an inactive instance's next sweep is not finite (the reference's mask blend makes its messages NaN); state restored, rerun the batch step-wise");
        return PDP_ERR_SPECULATION;
    }
    a->kernel_launches_host = 1; a->used_lds_host = 0; a->hbm_instances_host = p->B;
    a->iterations_run_host = (int)p->flags_host[FL_ITERS_RUN];
    a->decimator->has_prev = 1; p->has_edge_mask = 1;
    return PDP_OK;
}

static int sp_solve_speculative(pdp_problem *p, pdp_solve_args *a, void *stream);
extern "C" int pdp_sp_solve(pdp_problem *p, pdp_solve_args *a, void *stream)
{
    PDP_REQUIRE(p && a && p->av, "NULL argument / state not bound");
    PDP_REQUIRE(a->model == PDP_MODEL_SP || a->model == PDP_MODEL_REINFORCE, "persistent solve: SP and Reinforce triples only");
    PDP_REQUIRE(a->q && a->fs && a->active_mask && a->decimator, "NULL state array");
    if (p->exchange) {
        // one coupled forward over several processes (pdp_problem_set_exchange): the chunked LDS-resident loop completes its batch-wide
        // reductions across the parts; the other routes (lock-step launch, HBM-resident instances, step-wise fallback) are single-process
        PDP_REQUIRE(a->model == PDP_MODEL_SP && p->R == 1 && !a->isolate_instances, "coupled multi-process forward: the SP triple without batch replication");
        { const int st_ = resident_prepare(p); if (st_ != PDP_OK) return st_; }
        // Whether a part can take the resident loop is a LOCAL fact (an instance past the LDS limit, a clause-major edge order it does not
        // have), and the other parts are about to wait in the first chunk's exchange: the parts agree on it BEFORE anything else.  One part
        // that cannot -> every part returns PDP_ERR_SPECULATION (nothing was touched), which the host turns into "solve the segment whole
        // on one rank" (pdp.native.CoupledForwardFailed -> FactorGraphTrainerBase._predict_epoch).
        const size_t lds_x = lds2_bytes_for(p->res_fit_n, p->res_fit_m, p->res_fit_e);
        const bool can_resident = p->res_nbig == 0 && p->fn_edges_identity && p->res_nfit > 0 && lds_x <= 160 * 1024 - 1024 &&
                                  getenv("PDP_SOLVE_FORCE_HBM") == nullptr;
        {
            const uint32_t ors[1] = {can_resident ? 0u : 1u};
            uint32_t *m = nullptr;
            const int st_ = pdp_exchange_call(p, nullptr, 0, nullptr, 0, ors, 1, &m);
            if (st_ != PDP_OK) return st_;
            if (m[0]) {
                pdp_set_error(can_resident ? "coupled multi-process forward: another part cannot take the LDS-resident solver; solve the segment on one process"
                                           : "coupled multi-process forward: every instance of the part must fit the LDS-resident solver; solve the segment on one process");
                return PDP_ERR_SPECULATION;
            }
        }
        return sp_solve_speculative(p, a, stream);          // PDP_ERR_SPECULATION: the caller cannot rerun step-wise across processes
    }
    if (p->R > 1 && !a->replicas_identical) {
        // replicas that start from different states couple through the termination rule: lock-step, or not at all
        a->iterations_run_host = 0; a->used_lds_host = 0; a->kernel_launches_host = 0; a->replay_launches_host = 0; a->hbm_instances_host = 0;
        if (a->iterations <= 0) return PDP_OK;
        if (lockstep_possible(p, a)) return sp_solve_lockstep(p, a, ST(stream));
        pdp_set_error("persistent solve with batch replication: the replicas differ (they couple through the termination check) and the batch is too large "
                      "for the lock-step launch; run the batch step-wise");
        return PDP_ERR_SPECULATION;
    }
    if (getenv("PDP_SOLVE_FORCE_LOCKSTEP") && a->iterations > 0 && lockstep_possible(p, a)) return sp_solve_lockstep(p, a, ST(stream));      // (tests)
    const int rc = sp_solve_speculative(p, a, stream);
    if (rc == PDP_ERR_SPECULATION && lockstep_possible(p, a)) return sp_solve_lockstep(p, a, ST(stream));      // (every array is back at its call-entry state)
    return rc;
}

static int sp_solve_speculative(pdp_problem *p, pdp_solve_args *a, void *stream)
{
    PDP_REQUIRE(p && a && p->av, "NULL argument / state not bound");
    PDP_REQUIRE(a->model == PDP_MODEL_SP || a->model == PDP_MODEL_REINFORCE, "persistent solve: SP and Reinforce triples only");
    const bool rf_model = a->model == PDP_MODEL_REINFORCE;
    PDP_REQUIRE(!rf_model || a->coins, "persistent Reinforce needs the per-iteration coins (device array [iterations])");
    PDP_REQUIRE(!rf_model || !a->isolate_instances, "isolated-instance mode is implemented for the SP triple only");
    PDP_REQUIRE(a->q && a->fs && a->active_mask && a->decimator, "NULL state array");
    hipStream_t st = ST(stream);
    const int T = a->iterations;
    a->iterations_run_host = 0; a->used_lds_host = 0; a->kernel_launches_host = 0; a->replay_launches_host = 0; a->hbm_instances_host = 0;
    a->solve_kernel_ms_host = 0.0f; a->replay_kernel_ms_host = 0.0f;
    if (T <= 0) return PDP_OK;
    const size_t E = p->E, V = p->V, F = p->F, B = p->B;
    // The loop runs in chunks of C iterations (one launch each; the kernel resumes from the HBM state).  Chunking bounds
    // the cost of reproducing the reference's NaN poisoning: only the chunk in which the first NaN appears is partially
    // replayed, and every later chunk runs "poisoned from its first iteration" without any snapshot.
    // Chunk length: short chunks bound the poison replay, long ones amortise the launch, the record copies and the log pass of a launch's
    // first sweep.  Measured on MI355X (n=200, batch=5000, T=100, one NaN poison at sweep 81), ms per step: round 1-4 kernels 8 / 12 / 16 / 25
    // -> best at 12; round 5 (E2 takes the logs, later workgroups adopt the poison, so the replay is smaller and a launch's first sweep is
    // the expensive one): 12 / 14 / 16 / 18 / 20 / 25 -> 11.19 / 11.15 / 11.08 / 10.93 / 10.91 / 10.98; with pass 1 dispatched in the order of
    // the risk sort (the replay shrinks to a quarter): 20 / 25 / 27 / 34 / 40 / 50 -> 9.64 / 9.45 / 9.43 / 9.42 / 9.23 / 9.42 -- flat beyond 25
    // up to where the poison happens to sit in its chunk, and the order is a guess made at the end of the previous chunk, so not too long: 25.
    // A batch of ONE instance has nobody to supply the exact zero the speculation counts on (tools/spec_rate.py: it fails in the first
    // iteration for 10-19 of 20 random instances) -- but its batch-global minima are its own: the HBM-resident kernel computes them
    // (sp.exact), nothing is speculated, recorded or replayed, and the whole loop is one launch (a team of workgroups when the instance is big).
    // (the same holds for the R identical replicas of one instance: every replica's own minimum is the batch's)
    const bool exact = (B == 1 || (p->B0 == 1 && a->replicas_identical)) && !a->isolate_instances && getenv("PDP_SOLVE_NO_EXACT") == nullptr &&
                       !p->exchange;                       // (a part of one instance is not a batch of one)
    int C = 25;
    if (const char *env = getenv("PDP_SOLVE_CHUNK")) { const int v = atoi(env); if (v > 0) C = v; }
    if (C > T) C = T;
    { int st_ = resident_prepare(p); if (st_ != PDP_OK) return st_; }
    // Does the external-force column hold anything but zeros?  It does not for the p-d-p solver, so the call does not stop to find out: the
    // force-free instantiation is enqueued, k_solve_import -- which reads those cache lines anyway -- raises a flag if it meets a force, every
    // solver launch then returns before it touches anything, and the call runs once more with the force column (an extra 0.3 ms for the rare
    // caller with an external force, instead of a scan kernel and a host round trip in every call).
    for (int attempt = 0; attempt < 2; ++attempt) {
        const bool force_r = attempt == 1 || rf_model;
        // per-instance routing: the instances whose image fits run LDS-resident, the others on the HBM-resident kernel in the same chunk
        // loop; the launch is sized by the largest FITTING instance (Reinforce: the force is a 2-bit code in the slot word, no column)
        const int fn_ = p->res_fit_n, fm_ = p->res_fit_m, fe_ = p->res_fit_e;
        const size_t lds_r = lds2_bytes_for(fn_, fm_, fe_);
        const bool hbm_forced = getenv("PDP_SOLVE_FORCE_HBM") != nullptr;          // the switch lets the tests reach the HBM-resident kernel with small instances
        const bool mixed_ok = p->res_nbig == 0 || getenv("PDP_SOLVE_NO_ROUTING") == nullptr;
        const bool fits_r = p->fn_edges_identity && p->res_nfit > 0 && lds_r <= 160 * 1024 - 1024 && (!force_r || fn_ < 8192) && mixed_ok && !hbm_forced && !exact;
        // threads per instance: 256 for tiny instances, 512 while two workgroups share a CU, 1024 when the instance's LDS image allows
        // only one workgroup per CU (the same 16 waves per CU either way)
        int nt_r = fe_ <= 1024 ? 256 : (lds_r > 80 * 1024 ? 1024 : 512);
        if (const char *env = getenv("PDP_SOLVE_LDS_THREADS")) { const int v = atoi(env); if (v >= 64 && v <= 1024 && v % 64 == 0) nt_r = v; }
        if (!fits_r) break;
        a->hbm_instances_host = p->res_nbig;
        const int rc = sp_solve_resident(p, a, st, force_r, lds_r, nt_r, C);
        if (rc != PDP_RETRY_WITH_FORCE) return rc;
    }
    if (p->exchange) { pdp_set_error("coupled multi-process forward: the part did not take the LDS-resident loop"); return PDP_ERR_UNSUPPORTED; }
    a->hbm_instances_host = p->B;
    // ---- instances too large for the LDS: HBM-resident kernel, host-driven chunk loop -----------------------------------

    // one allocation: speculation record [2C] + control words + per-instance records + two snapshots (call entry, chunk entry)
    // (HBM-resident kernel only; the LDS path keeps its state in the instance records and needs the call-entry snapshot alone)
    const size_t words = 2 * (size_t)C + 8;
    const size_t snap_floats = 3 * E + 2 * E + V + F + V + B + E + E + B;
    const size_t snap_bytes = snap_floats * 4 + ((B + 63) & ~(size_t)63);
    const size_t blob_bytes = words * 4 + 2 * B * 4 + 2 * snap_bytes + 64;
    if (p->solve_blob_bytes < blob_bytes) {
        if (p->solve_blob) pdp_dev_free(p->solve_blob);
        p->solve_blob = nullptr; p->solve_blob_bytes = 0;
        { int st_ = pdp_dev_alloc((void **)&p->solve_blob, blob_bytes); if (st_ != PDP_OK) return st_; }
        p->solve_blob_bytes = blob_bytes;
    }
    if (p->solve_host_words < words) {
        if (p->solve_host) (void)hipHostFree(p->solve_host);
        p->solve_host = nullptr; p->solve_host_words = 0;
        PDP_HIP_CHECK(hipHostMalloc((void **)&p->solve_host, words * 4));
        p->solve_host_words = words;
    }
    char *blob = p->solve_blob;
    uint32_t *spec = (uint32_t *)blob;
    uint32_t *ctl = spec + 2 * (size_t)C;            // [0] nan_iter, [1] force flag, [2] replay count
    int32_t *last_event = (int32_t *)(blob + words * 4);
    int32_t *replay_list = last_event + B;
    auto carve_snap = [&](char *base) {
        SolveSnapshot sn; float *f = (float *)base;
        sn.q = f; f += 3 * E; sn.fs = f; f += 2 * E; sn.av = f; f += V; sn.af = f; f += F; sn.sol = f; f += V;
        sn.sat = f; f += B; sn.emask = f; f += E; sn.prev = f; f += E; sn.cnt = f; f += B; sn.amask = (uint8_t *)f;
        return sn;
    };
    SolveSnapshot snap0 = carve_snap((char *)(replay_list + B));            // state at call entry (speculation failure)
    SolveSnapshot snap = carve_snap((char *)(replay_list + B) + snap_bytes); // state at chunk entry (poison replay)
    const int had_prev0 = a->decimator->has_prev, had_emask0 = p->has_edge_mask;
    int status = snapshot_copy(p, a, snap0, true, st);
    if (status != PDP_OK) return status;

    SolveParams sp;
    memset(&sp, 0, sizeof(sp));
    sp.tol = a->tolerance; sp.t_max = a->t_max; sp.pi = a->pi;
    sp.q = a->q; sp.fs = a->fs; sp.amask = a->active_mask;
    sp.prev = a->decimator->prev; sp.counters = a->decimator->counters;
    sp.check_termination = a->check_termination;
    sp.spec_used = spec; sp.spec_zero = spec + C; sp.nan_iter = ctl;
    sp.w_perm_zero = p->flags + FL_PERM_ZERO; sp.w_iters_run = p->flags + FL_ITERS_RUN; sp.w_violation = p->flags + FL_SPEC_VIOLATION;
    sp.last_event = last_event;
    sp.rf = rf_model ? 1 : 0; sp.isolate = a->isolate_instances ? 1 : 0; sp.coins = a->coins; sp.dprob = a->decimation_probability;

    PDP_HIP_CHECK(hipMemsetAsync(ctl, 0, sizeof(uint32_t) * 8, st));
    const bool fits = false;
    const size_t lds = 0;
    float *extra_v = nullptr;
    { const int st_ = hbm_workspaces(p, sp); if (st_ != PDP_OK) return st_; }
    extra_v = p->solve_extra_v;
    if (exact) {
        sp.exact = 1; sp.T = T; sp.has_prev = a->decimator->has_prev; sp.has_edge_mask = p->has_edge_mask; sp.final_chunk = 1; sp.poison_from = 0x7fffffff;
        PDP_HIP_CHECK(hipMemsetAsync(ctl, 0xff, sizeof(uint32_t), st));
        PDP_HIP_CHECK(hipMemsetAsync(p->flags + FL_PERM_ZERO, 0xff, sizeof(uint32_t), st));
        PDP_HIP_CHECK(hipMemsetAsync(p->flags + FL_SPEC_VIOLATION, 0, sizeof(uint32_t) * 2, st));
        { const int st_ = launch_hbm(p, sp, p->B, st, true); if (st_ != PDP_OK) return st_; }
        PDP_LAUNCH_CHECK();
        PDP_HIP_CHECK(hipMemcpyAsync(p->flags_host, p->flags, FL_COUNT * sizeof(uint32_t), hipMemcpyDeviceToHost, st));
        PDP_HIP_CHECK(hipStreamSynchronize(st));
        if (p->flags_host[FL_SPEC_VIOLATION] != 0u) {
            // the instance went inactive in a state whose next sweep is not finite: the reference's mask blend turns that into NaN messages
            // (see ghost_bad in k_sp_solve); the strict step-wise loop evaluates it literally
            status = snapshot_copy(p, a, snap0, false, st);
            if (status == PDP_OK) status = hipStreamSynchronize(st) == hipSuccess ? PDP_OK : PDP_ERR_HIP;
            if (status != PDP_OK) return status;
            pdp_set_error("persistent solve: an inactive instance's next sweep is not finite (the reference's mask blend makes its messages NaN); state restored, rerun the batch step-wise");
            return PDP_ERR_SPECULATION;
        }
        a->kernel_launches_host = 1;
        a->iterations_run_host = (int)p->flags_host[FL_ITERS_RUN];
        a->decimator->has_prev = 1; p->has_edge_mask = 1;
        return PDP_OK;
    }
    auto set_src_live = [&]() {
        sp.inst_list = nullptr;
        sp.src_q = a->q; sp.src_fs = a->fs; sp.src_av = p->av; sp.src_af = p->af; sp.src_sol = p->sol; sp.src_sat = p->is_sat;
        sp.src_emask = p->emask; sp.src_prev = a->decimator->prev; sp.src_cnt = a->decimator->counters; sp.src_amask = a->active_mask;
    };
    uint32_t *host = p->solve_host;
    const bool debug = getenv("PDP_DEBUG") != nullptr;
    bool ok = true, poisoned_all = false;
    int done = 0, total_iters = 0;
    while (status == PDP_OK && ok && done < T) {
        const int c = (T - done) < C ? (T - done) : C;
        sp.T = c; sp.has_prev = a->decimator->has_prev; sp.has_edge_mask = p->has_edge_mask; sp.chunk_start = done;
        sp.final_chunk = (done + c >= T) ? 1 : 0;
        set_src_live();
        int poison_from = poisoned_all ? 0 : 0x7fffffff;
        uint32_t n_replayed = 0;
        if (!poisoned_all) { status = snapshot_copy(p, a, snap, true, st); if (status != PDP_OK) break; }
        for (int pass = 0; pass < 2; ++pass) {
            sp.poison_from = poison_from;
            if (pass == 0 || !fits) {
                PDP_HIP_CHECK(hipMemsetAsync(spec, 0, sizeof(uint32_t) * 2 * (size_t)C, st));
                PDP_HIP_CHECK(hipMemsetAsync(ctl, 0xff, sizeof(uint32_t), st));
                PDP_HIP_CHECK(hipMemsetAsync(ctl + 2, 0, sizeof(uint32_t), st));
                PDP_HIP_CHECK(hipMemsetAsync(p->flags + FL_PERM_ZERO, 0xff, sizeof(uint32_t), st));
            }
            // violation (+ iters_run, except for the selective replay which extends pass 1's maximum)
            PDP_HIP_CHECK(hipMemsetAsync(p->flags + FL_SPEC_VIOLATION, 0, sizeof(uint32_t) * (sp.inst_list ? 1 : 2), st));
            status = launch_hbm(p, sp, p->B, st, true);
            if (status != PDP_OK) break;
            PDP_LAUNCH_CHECK();
            a->kernel_launches_host++;
            PDP_HIP_CHECK(hipMemcpyAsync(host, spec, words * 4, hipMemcpyDeviceToHost, st));
            PDP_HIP_CHECK(hipMemcpyAsync(p->flags_host, p->flags, FL_COUNT * sizeof(uint32_t), hipMemcpyDeviceToHost, st));
            PDP_HIP_CHECK(hipStreamSynchronize(st));
            const uint32_t t_nan = host[2 * (size_t)C];
            if (pass == 0 && !poisoned_all && t_nan < (uint32_t)c && !a->isolate_instances) {
                // Some instance poisons the batch from (chunk-relative) iteration t_nan on.
                poison_from = (int)t_nan;
                poisoned_all = true;
                if (p->flags_host[FL_SPEC_VIOLATION]) { ok = false; break; }
                {
                    status = snapshot_copy(p, a, snap, false, st);       // HBM kernel works in place: full replay of the chunk
                    if (status != PDP_OK) break;
                }
                continue;
            }
            break;
        }
        if (status != PDP_OK || !ok) break;
        ok = p->flags_host[FL_SPEC_VIOLATION] == 0u;
        const uint32_t perm_from = p->flags_host[FL_PERM_ZERO];
        for (int t = 0; t < c && t < poison_from && ok && !a->isolate_instances; ++t)
            if ((uint32_t)t < perm_from && (host[t] & ~host[C + t]) != 0u) ok = false;
        if (debug)
            fprintf(stderr, "[pdp_sp_solve] chunk@%d len=%d violation=%u perm_from=%u poison_from=%d replayed=%u iters=%u lds=%zu ok=%d\n", done, c,
                    p->flags_host[FL_SPEC_VIOLATION], perm_from, poison_from, n_replayed, p->flags_host[FL_ITERS_RUN], lds, (int)ok);
        if (!ok) break;
        const int it = (int)p->flags_host[FL_ITERS_RUN];
        total_iters = done + it;
        a->decimator->has_prev = 1;
        p->has_edge_mask = 1;
        done += c;
        if (it < c) break;                // every instance went inactive inside this chunk (global early exit, solver.py:383)
    }
    if (status == PDP_OK && !ok) {
        // leave the caller's state exactly as it was at call entry so that it can rerun the batch step-wise
        status = snapshot_copy(p, a, snap0, false, st);
        if (status == PDP_OK) status = hipStreamSynchronize(st) == hipSuccess ? PDP_OK : PDP_ERR_HIP;
        a->decimator->has_prev = had_prev0; p->has_edge_mask = had_emask0;
    }
    (void)extra_v;
    if (status != PDP_OK) return status;
    if (!ok) {
        pdp_set_error("persistent solve: a cross-instance coupling of the reference became active (batch-global min != 0); "
                      "state restored, rerun the batch step-wise");
        return PDP_ERR_SPECULATION;
    }
    a->iterations_run_host = total_iters;
    return PDP_OK;
}
