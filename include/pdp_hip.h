/*
 * pdp_hip.h -- C ABI of the MI355X-native PDP hot path (libpdp_hip.so).
 *
 * The reference (microsoft/PDP-Solver) has no FFI: its hot path is Python calling torch operators.
 * This header is therefore the boundary a maintainer binds (ctypes stub in INTEGRATION.md) in place
 * of those operator call-site groups.  Every entry point names the reference code it replaces.
 * All pointers are DEVICE pointers unless the name ends in `_host`; all arrays are dense,
 * C-contiguous, fp32 / int32 / uint8 / int64 as declared; `stream` is a hipStream_t (may be NULL).
 * Functions return PDP_OK (0) or an error code; pdp_last_error() gives the message.  Unless stated
 * otherwise calls are asynchronous on `stream`.
 *
 * Batch layout (reference: src/pdp/factorgraph/dataset.py:165-187): instance ids in
 * batch_variable_map / batch_function_map are non-decreasing, variable / clause ids are global and
 * contiguous per instance, edges of one instance are contiguous.  Anything else -> PDP_ERR_LAYOUT.
 */
#ifndef PDP_HIP_H
#define PDP_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* 2 (round 5): pdp_train_gru_backward's scratch contract and the training entry points added in round 4; a coupled multi-process forward
 * (pdp_problem_set_exchange) reports PDP_ERR_SPECULATION on EVERY part when one part cannot take the resident loops */
#define PDP_ABI_VERSION 3

enum {
    PDP_OK = 0,
    PDP_ERR_INVALID = 1,      /* bad argument */
    PDP_ERR_LAYOUT = 2,       /* batch not in the loader's instance-contiguous layout */
    PDP_ERR_HIP = 3,          /* HIP runtime error (no device, OOM, launch failure) */
    PDP_ERR_UNSUPPORTED = 4,
    PDP_ERR_SPECULATION = 5   /* persistent solve hit a cross-instance coupling; rerun step-wise */
};

enum { PDP_RNG_STREAM = 0, PDP_RNG_PHILOX = 1 };
enum { PDP_MODEL_SP = 0, PDP_MODEL_WALKSAT = 1, PDP_MODEL_REINFORCE = 2 };

typedef struct pdp_problem pdp_problem;       /* a batch of CNF instances resident in HBM */
typedef struct pdp_decimator pdp_decimator;   /* SequentialDecimator / ReinforceDecimator state */

int pdp_abi_version(void);
const char *pdp_last_error(void);
int pdp_device_count(void);

/* ---- problem container --------------------------------------------------------------------
 * replaces: SATProblem.__init__/setup_problem, the 24 sparse COO masks and _replicate_batch
 * (reference: src/pdp/nn/solver.py:22-178).  graph_map is [2,E] (row 0 variable id, row 1 clause
 * id), edge_feature [E] in {-1,+1}.  With replication R > 1 the handle holds R copies laid out as
 * solver.py:56-82 does (replica r of instance i has id i + r*B).  Synchronous. */
int pdp_problem_create(pdp_problem **out, int E, int V, int F, int B, int replication,
                       const int32_t *graph_map, const int32_t *batch_variable_map,
                       const int32_t *batch_function_map, const float *edge_feature, void *stream);
int pdp_problem_destroy(pdp_problem *p);
/* dims_host[8] = {E, V, F, B, R, max_vars, max_clauses, max_edges} of the (replicated) batch */
int pdp_problem_dims(const pdp_problem *p, int32_t *dims_host);
/* replicated graph arrays for the Python-visible attributes (_graph_map ...); any may be NULL */
int pdp_problem_export_graph(const pdp_problem *p, int32_t *graph_map, int32_t *batch_variable_map,
                             int32_t *batch_function_map, float *edge_feature, void *stream);
/* binds caller-owned state tensors: _active_variables [V], _active_functions [F], _solution [V],
 * _is_sat [B], _edge_mask [E] (reference: solver.py:49-54,370) and initialises them (1, 1, 0.5, 0.5, 1) */
int pdp_problem_bind_state(pdp_problem *p, float *active_variables, float *active_functions,
                           float *solution, float *is_sat, float *edge_mask, void *stream);

/* ---- K7: simplification ---------------------------------------------------------------------
 * replaces: SATProblem.simplify/_propagate_single_clauses/_peel (solver.py:180-203,228-285) */
int pdp_simplify(pdp_problem *p, void *stream);
/* replaces: SATProblem.set_variables/_set_variable_core (solver.py:205-226,275-279);
 * assignment [V] in {-1,0,+1} is masked in place by the active flags like the reference does */
int pdp_set_variables(pdp_problem *p, float *assignment, void *stream);
/* K8 replaces: solver.py:370-371 / 439-440.  Writes the bound _edge_mask; if all_active_host is
 * non-NULL the call synchronises and stores 1 when every edge is still active, else 0. */
int pdp_refresh_edge_mask(pdp_problem *p, int32_t *all_active_host, void *stream);

/* ---- K4 / K5 ----------------------------------------------------------------------------------
 * replaces: util.sparse_smooth_max (util.py:282-286) with mask = variable mask: x [E] -> out [V] */
int pdp_smooth_max(pdp_problem *p, const float *x, float *out, void *stream);
/* replaces: util.sparse_max / sparse_argmax (util.py:257-275) with mask = batch-variable mask,
 * including the batch-global `x - x.min() + 1` shift: x [V] -> out [B] */
int pdp_instance_max(pdp_problem *p, const float *x, float *out, void *stream);
int pdp_instance_argmax(pdp_problem *p, const float *x, int64_t *out, void *stream);

/* ---- K1-K3: survey propagation sweep ------------------------------------------------------------
 * replaces: SurveyPropagator.forward (pdp_propagate.py:139-221, include_adaptors=False).
 * dec_q [E,3], dec_fs [E,2] = decimator_state; edge_mask [E] or NULL; active_mask uint8 [B] or NULL;
 * init_q / init_fs = init_state; outputs out_q [E,3], out_fs [E,2] (may not alias the inputs). */
int pdp_sp_propagate(pdp_problem *p, const float *dec_q, const float *dec_fs, const float *edge_mask,
                     const uint8_t *active_mask, const float *init_q, const float *init_fs, float pi,
                     float *out_q, float *out_fs, void *stream);

/* ---- adaptor form of the propagator (model type p-nd-np) ------------------------------------------
 * replaces: the include_adaptors=True branches of SurveyPropagator.forward (pdp_propagate.py:166-167, 179-182).
 * dec_v / dec_f [E,H] = neural decimator state; w_f [H] = _function_input_projector.weight, W_v [2,H] =
 * _variable_input_projector.weight; outputs xlog [E] = logsigmoid(w_f . dec_v) and fs2 [E,2] =
 * (sigmoid(W_v[0] . dec_f), sign(W_v[1] . dec_f)). */
int pdp_sp_adaptors(pdp_problem *p, int H, const float *dec_v, const float *dec_f, const float *w_f, const float *W_v,
                    float *xlog, float *fs2, void *stream);
/* pdp_sp_propagate with the clause-to-variable input already in the log domain (xlog [E] from pdp_sp_adaptors) */
int pdp_sp_propagate_adapted(pdp_problem *p, const float *xlog, const float *dec_fs, const float *edge_mask,
                             const uint8_t *active_mask, const float *init_q, const float *init_fs, float pi,
                             float *out_q, float *out_fs, void *stream);
/* K6 replaces: SurveyScorer.forward (pdp_predict.py:155-192): fs [E,2] -> score [V] */
int pdp_survey_score(pdp_problem *p, const float *fs, float pi, float *score, void *stream);

/* ---- K9 / K13 -------------------------------------------------------------------------------------
 * replaces: SatCNFEvaluator.forward (util.py:210-236): pred [V] -> solved [B], unsat_clauses [B] */
int pdp_cnf_eval(pdp_problem *p, const float *pred, float *solved, float *unsat, void *stream);

/* replaces: SatLossEvaluator.forward (util.py:178-197), the energy loss of a prediction reported by the test mode
 * (trainer.py:108-123, base.py:223-250).  pred [V]; coeff = min(global_step^exploration, max_coeff) (host); loss_sharpness a
 * positive integer; loss: device float [1] = mean over the clauses of the batch. */
int pdp_sat_loss(pdp_problem *p, const float *pred, float coeff, float eps, int sharpness, float *loss, void *stream);
/* replaces: PropagatorDecimatorSolverBase._update_solution (solver.py:388-399): out [V] */
int pdp_update_solution(pdp_problem *p, const float *pred, float *out, void *stream);
/* replaces: SatFactorGraphTrainer._check_recurrence_termination (trainer.py:150-162), replication aware */
int pdp_check_termination(pdp_problem *p, uint8_t *active_mask, const float *pred, void *stream);
/* replaces: the per-sweep host read that ends the plug-in loop, `if int(active_mask.sum()) <= 0: break` (solver.py:383-384), for callers that
 * enqueue the sweeps of a forward without waiting for them (the neural triples: one captured HIP graph per sweep parity, pdp/nn/solver.py).
 * pdp_loop_begin clears the loop's two device words; pdp_loop_step, enqueued behind a sweep's termination check, counts the sweep and raises
 * the stop word when no instance of active_mask [B] is active any more (active_mask NULL: no termination check, it only counts); behind
 * the stop word the operators that write message states under an active mask (pdp_neural_aggregate_edges, pdp_neural_gru,
 * pdp_sp_propagate / _adapted) enqueue kernels that return at once, so sweeps already in flight change nothing; pdp_loop_read waits for
 * the stream and returns the stop word and the executed sweeps (= the reference's iteration count), and with end != 0 clears both words. */
int pdp_loop_begin(pdp_problem *p, void *stream);
int pdp_loop_step(pdp_problem *p, const uint8_t *active_mask, void *stream);
int pdp_loop_read(pdp_problem *p, int32_t *stopped_host, int32_t *iterations_host, int end, void *stream);

/* ---- decimators --------------------------------------------------------------------------------------
 * replaces: SequentialDecimator state (_previous_function_state, _counters; pdp_decimate.py:115-125,179-183) */
int pdp_decimator_create(pdp_decimator **out, pdp_problem *p);
int pdp_decimator_destroy(pdp_decimator *d);
int pdp_decimator_reset(pdp_decimator *d, void *stream);
/* replaces: SequentialDecimator.forward with scorer = SurveyScorer (pdp_decimate.py:122-177) */
int pdp_sequential_decimate(pdp_problem *p, pdp_decimator *d, const float *fs, uint8_t *active_mask,
                            float tolerance, float t_max, float pi, void *stream);
/* the same in two halves for a non-native scorer plug-in: gate = :124-150 (returns via
 * any_converged_host, synchronises), apply = :156-173 with a caller-computed score [V] */
int pdp_sequential_decimate_gate(pdp_problem *p, pdp_decimator *d, const float *fs, uint8_t *active_mask,
                                 float tolerance, float t_max, int32_t *any_converged_host, void *stream);
int pdp_sequential_decimate_apply(pdp_problem *p, pdp_decimator *d, const float *fs, const float *score,
                                  const uint8_t *active_mask, void *stream);
/* replaces: ReinforceDecimator.forward (pdp_decimate.py:202-234); fs [E,2] updated in place; coin =
 * the caller's torch.rand(1) draw */
int pdp_reinforce_decimate(pdp_problem *p, pdp_decimator *d, float *fs, uint8_t *active_mask, float coin,
                           float decimation_probability, float pi, void *stream);
/* replaces: ReinforcePredictor.forward (pdp_predict.py:221-226): fs [E,2] -> pred [V] */
int pdp_reinforce_predict(pdp_problem *p, const float *fs, float *pred, void *stream);

/* ---- K14: Walk-SAT ------------------------------------------------------------------------------------
 * replaces: _compute_energy (solver.py:486-496): assignment [V] -> energy [B], unsat_functions [F] */
int pdp_energy(pdp_problem *p, const float *assignment, float *energy, float *unsat_functions, void *stream);
/* replaces: _compute_energy_diff (solver.py:469-484) using the bound _edge_mask: -> delta [V] */
int pdp_energy_diff(pdp_problem *p, const float *assignment, float *delta, void *stream);
/* The in-kernel (PDP_RNG_PHILOX) draws of pdp_random_fill / pdp_local_search are Philox counters of the variable's / the instance's index
 * inside the forward (pdp_predict.py:125-126 draws rand(n_active), solver.py:457,460 rand(V,1) and rand(B): one number per variable /
 * instance of the batch).  When the batch of this problem is a contiguous PART of a larger forward solved elsewhere (isolated instances
 * dealt to several GPUs, pdp/parallel.py) the caller states where the part starts: the draws are then those of the whole forward.
 * Replication 1 only.  Default 0, 0. */
int pdp_problem_set_rng_base(pdp_problem *p, uint32_t first_variable, uint32_t first_instance);
/* One COUPLED forward solved by several processes (one per GPU), each holding a contiguous part of the batch (pdp_problem_set_rng_base says
 * where).  The reference couples the instances of a forward through batch-wide reductions -- the first sweep with a NaN survey (it stops the
 * decimation of the whole batch), the exact zero inside util.sparse_max / sparse_argmax's x - min(x) + 1, the executed sweeps, and the same
 * zero in every Walk-SAT step.  The chunked LDS-resident solver (pdp_sp_solve) and the persistent Walk-SAT
 * (pdp_local_search) take these from a few control words per chunk; with a callback set they hand those words to it between the launches and
 * continue with what it returns: fn(user, mins, n_mins, maxs, n_maxs, ors, n_ors) replaces every element by its minimum / maximum /
 * bit-wise OR over all parts, in place (host memory), and returns 0.  Every part calls it the same number of times (the control flow depends
 * on the merged words only).  Restrictions: SP triple, no batch replication, every instance of a part fits the LDS-resident solver; a failed
 * speculation cannot fall back to the step-wise loop across processes: PDP_ERR_SPECULATION / PDP_ERR_UNSUPPORTED go to the caller.
 * NULL removes the callback. */
typedef int (*pdp_exchange_fn)(void *user, uint32_t *mins, int n_mins, uint32_t *maxs, int n_maxs, uint32_t *ors, int n_ors);
int pdp_problem_set_exchange(pdp_problem *p, pdp_exchange_fn fn, void *user);
/* replaces: IdentityPredictor.forward(last_call=True) random fill (pdp_predict.py:121-126).
 * PDP_RNG_STREAM: values [n_active] are consumed in variable order; PDP_RNG_PHILOX: in-kernel. */
int pdp_random_fill(pdp_problem *p, int rng_mode, const float *values, uint64_t seed, void *stream);
/* replaces: PropagatorDecimatorSolverBase._local_search (solver.py:433-467).  pred [V] -> out [V].
 * PDP_RNG_STREAM: var_rand [iterations,V] and coin_rand [iterations,B] hold the torch.rand draws of
 * each step.  Synchronises; steps_host receives the number of executed steps (global early exit).
 * All steps run in one persistent launch per instance (LDS-resident, or an HBM-resident form for instances past
 * the LDS limit, routed per instance); the strict three-launches-per-step loop is the fallback. */
int pdp_local_search(pdp_problem *p, const float *pred, int iterations, float epsilon, int rng_mode,
                     const float *var_rand, const float *coin_rand, uint64_t seed, float *out,
                     int32_t *steps_host, void *stream);
/* replaces: _deduplicate (solver.py:401-431, with the integer-division fix): pred [V] -> out [V/R],
 * chosen replica per original instance [B/R] (may be NULL) */
int pdp_deduplicate(pdp_problem *p, const float *pred, float *out, int32_t *chosen, void *stream);

/* ---- persistent solve: the whole _forward_core loop in one launch ------------------------------------
 * replaces: PropagatorDecimatorSolverBase._forward_core (solver.py:355-386) for the classical
 * triples (SurveyPropagator + SequentialDecimator + IdentityPredictor, or the Reinforce triple),
 * one workgroup per instance with the instance resident in LDS; instances past the LDS limit run
 * on an HBM-resident kernel inside the same call (per-instance routing; a big instance is spread
 * over a team of workgroups).  The kernels assume the reference's accidental cross-instance
 * couplings are inert (batch-global min == 0); a NaN survey, which poisons the whole batch in the
 * reference, is detected and the affected instances are replayed on the device.  If a coupling was
 * active the call restores every array it touched and reruns a batch of up to 1 024 instances in a
 * lock-step launch that exchanges the couplings exactly; a larger batch returns PDP_ERR_SPECULATION:
 * the caller reruns it through the step-wise entry points above.  A batch of ONE instance (or the
 * identical replicas of one) is solved exactly instead -- its batch-global minima are its own --
 * and never fails.  Synchronises. */
typedef struct pdp_solve_args {
    int32_t model;                /* PDP_MODEL_SP or PDP_MODEL_REINFORCE */
    int32_t iterations;           /* T */
    float tolerance, t_max, pi, decimation_probability;
    uint64_t seed;                /* reserved */
    const float *coins;           /* PDP_MODEL_REINFORCE: device array [iterations], the shared coin torch.rand(1) of every iteration
                                   * (pdp_decimate.py:218) drawn by the caller in order; iterations_run_host says how many were consumed */
    float *q;                     /* [E,3] in: init propagator state[0], out: final */
    float *fs;                    /* [E,2] in/out */
    uint8_t *active_mask;         /* [B] in/out */
    pdp_decimator *decimator;     /* in/out */
    int32_t check_termination;    /* 1: per-iteration CNF check de-activates solved instances */
    int32_t iterations_run_host;  /* out: executed iterations (max over instances) */
    int32_t used_lds_host;        /* out: 1 if the LDS-resident variant ran */
    int32_t kernel_launches_host; /* out: launches of the solver kernel that did work (one per chunk of iterations) */
    int32_t replay_launches_host; /* out: poison-replay launches that did work */
    int32_t time_kernels;         /* in: 1 = bracket every chunk launch with HIP events and report the sums below */
    float solve_kernel_ms_host;   /* out: device time of the chunk launches (time_kernels == 1) */
    float replay_kernel_ms_host;  /* out: device time from the end of a chunk launch that was replayed to the start of the next chunk's: the replay
                                   * launch and the four control kernels around it (time_kernels == 1) */
    int32_t replicas_identical;   /* in: 1 = the R replicas of every instance (batch replication, solver.py:56-82) start from identical state,
                                   * so their trajectories coincide and the replica-aware termination rule (trainer.py:157-160) equals the
                                   * per-replica one.  0 with replication > 1 (random initial state): the replicas couple through the
                                   * termination rule -- batches of up to 1 024 instances run in the lock-step launch, larger ones return
                                   * PDP_ERR_SPECULATION untouched */
    int32_t isolate_instances;    /* in: 1 = "fixed" semantics instead of the reference's: every instance is solved on its own -- the
                                   * batch-global minimum of sparse_max / sparse_argmax is taken as 0 and a NaN survey stops the decimation
                                   * of its own instance only (in the reference it stops the whole batch).  Never
                                   * returns PDP_ERR_SPECULATION. */
    int32_t hbm_instances_host;   /* out: instances of the batch that did not fit the LDS and ran on the HBM-resident kernel inside the same chunk
                                   * loop (per-instance routing; 0 when every instance fits, the batch size when none does) */
    int32_t inputs_disposable;    /* in: 1 = the caller does not need q / fs back when the call returns PDP_ERR_SPECULATION (it still holds the
                                   * state it built them from, as PropagatorDecimatorSolverBase does: its init_state is never mutated,
                                   * solver.py:355-365) -- the call then skips their call-entry snapshot (250 MB of copies on config 2).  Every
                                   * other array is restored as always; 0 keeps the full guarantee */
} pdp_solve_args;
int pdp_sp_solve(pdp_problem *p, pdp_solve_args *args, void *stream);

/* ---- DIMACS ingestion (host code, no GPU needed) -----------------------------------------------------------
 * Native twin of the reference's converter front end (reference: src/dimacs2json.py:22-51 parsing, :43-51,85-91
 * compaction; consumed by src/pdp/factorgraph/dataset.py:120-136).  pdp_dimacs_open parses one file in a single pass and
 * reports the sizes of the COMPACT instance (unused variables dropped, empty clauses dropped, last occurrence of a
 * variable inside a clause wins); pdp_dimacs_read copies the edge list: signed_vars[e] = +-(1-based compact variable id),
 * clause_ids[e] = 1-based clause id, clause-major with ascending variable index -- the second and third list of the
 * reference's JSON line.  Malformed input -> PDP_ERR_INVALID with file:line in pdp_last_error(). */
typedef struct pdp_dimacs pdp_dimacs;
int pdp_dimacs_open(const char *path, pdp_dimacs **out, int32_t *n_vars, int32_t *n_clauses, int64_t *n_edges);
int pdp_dimacs_read(const pdp_dimacs *d, int32_t *signed_vars /*[n_edges]*/, int32_t *clause_ids /*[n_edges]*/);
int pdp_dimacs_close(pdp_dimacs *d);
/* count files parsed by `threads` host threads; out / n_vars / n_clauses / n_edges are arrays of `count` entries.  On a failure every
 * handle is released and the first error is reported. */
int pdp_dimacs_open_many(const char *const *paths, int32_t count, int32_t threads, pdp_dimacs **out, int32_t *n_vars,
                         int32_t *n_clauses, int64_t *n_edges);

/* ---- neural plug-ins: per-edge MLP / GRU layers on the fp32 matrix cores ---------------------------------
 * Weights are handed over PRE-TRANSPOSED and ZERO PADDED by the host: a layer y = act(W x + b) with nn.Linear weight
 * W [N, K] is passed as Wt [Kp, Np] row-major with Wt[k, j] = W[j, k], Kp = K rounded up to even, Np = N rounded up
 * to a multiple of 32, biases padded to Np with zeros.  Products are k-ordered fp32 fma chains starting at the bias. */
typedef struct pdp_agg_desc {      /* MessageAggregator (reference: src/pdp/nn/util.py:11-77) */
    const float *Wt1m, *b1m, *Wt2m, *Wt1a, *b1a, *Wt2a;   /* W1_m (+bias), W2_m, W1_a (+bias), W2_a */
    int32_t din;                    /* input width = state width + 1 (edge sign is appended by the kernel) */
    int32_t m1, a, g, out;          /* mem_hidden, mem_agg_hidden, agg_hidden, output widths */
    int32_t fd;                     /* 1: edge sign appended after aggregation (include_self_message=False), else 0 */
} pdp_agg_desc;
typedef struct pdp_gru_desc {      /* nn.GRUCell: Wt_ih [Kp(dx+1), 3*Hp], Wt_hh [Kp(H), 3*Hp], gate g at columns g*Hp.. */
    const float *Wt_ih, *Wt_hh, *b_ih, *b_hh;
    int32_t dx, H;
} pdp_gru_desc;
typedef struct pdp_head_desc {     /* Perceptron / PerceptronTanh head: Wt1 [Kp(H), Np(C)], b1 [Np], w2 [C] */
    const float *Wt1, *b1, *w2;
    int32_t H, C, out_act;         /* out_act: 3 sigmoid (trainer.py:28-29), 4 tanh (util.py:250-251) */
} pdp_head_desc;
/* replaces: one MessageAggregator call of NeuralMessagePasser.forward (pdp_propagate.py:77-78 by_variable=1, :88-89
 * by_variable=0): out [E,out] = mask * Agg([state ‖ sign]) + (1 - mask) * old, mask from active_mask (NULL: ones) */
int pdp_neural_aggregate_edges(pdp_problem *p, const pdp_agg_desc *d, int by_variable, const float *state,
                               const float *edge_mask, const uint8_t *active_mask, const float *old, float *out, void *stream);
/* replaces: one GRU direction of NeuralDecimator.forward (pdp_decimate.py:70-75, 78-83): out [E,H] (must not alias h) */
int pdp_neural_gru(pdp_problem *p, const pdp_gru_desc *d, const float *state, const float *h, const uint8_t *active_mask,
                   float *out, void *stream);
/* replaces: NeuralPredictor.forward variable branch (pdp_predict.py:67-77): state [E,H] -> pred [V] */
int pdp_neural_predict(pdp_problem *p, const pdp_agg_desc *d, const pdp_head_desc *head, const float *state,
                       const float *edge_mask, float *pred, void *stream);

/* ---- training path (the gradients loss.backward() needs in FactorGraphTrainerBase._train_batch, src/pdp/factorgraph/base.py:149-182) ------------
 * The reference differentiates its torch operators with autograd.  Each differentiable building block of the np-nd-np solver has a
 * forward and an adjoint entry point here; the host wraps a pair in an autograd node (pdp/nn/train_ops.py) and keeps the graph bookkeeping,
 * gradient clipping and the caller's optimizer.  Activations are identified like pdp_head_desc.out_act: 0 none, 1 logsigmoid, 2 relu,
 * 3 sigmoid, 4 tanh; activation derivatives are taken from the saved OUTPUT.  All matrices row-major fp32. */
/* replaces: nn.Linear + activation (util.py:56,74 MessageAggregator layers; trainer.py:28-29 Perceptron): Y [R,N] = act(X [R,K] W^T + b), W [N,K] */
int pdp_train_linear(const float *X, int64_t R, int K, int64_t ldx, const float *W, const float *b, int N, int act, float *Y, void *stream);
/* adjoint: dZ = dY * act'(Y) (dZ [R,N]: scratch, may alias dY); dX [R,K] = dZ W (NULL: skipped); dW [N,K] = dZ^T X; db [N] = column sums (NULL: no bias) */
int pdp_train_linear_backward(const float *dY, const float *Y, const float *X, int64_t R, int K, int64_t ldx, const float *W, int N, int act,
                              float *dZ, float *dX, int64_t lddx, float *dW, float *db, void *stream);
/* The same layer on an operand whose LAST column is stored apart -- Y = act([X | xs] W^T + b) with X [R,K], xs [R] (the edge sign the reference
 * concatenates in front of every layer of the training path, pdp_propagate.py:66-67, util.py:71-72), W [N, K + 1]: the K-wide block runs on the
 * row-stripe GEMM, the last column is a rank-one term of its epilogue; no [R, K + 1] copy is made.  Only for shapes that kernel takes
 * (pdp_train_linear_s_supported != 0: >= 4 096 rows, act none / logsigmoid, the weight chunk fits the LDS); otherwise concatenate. */
int pdp_train_linear_s_supported(int64_t R, int K, int N, int act);
int pdp_train_linear_s(const float *X, const float *xs, int64_t R, int K, int64_t ldx, const float *W, const float *b, int N, int act, float *Y, void *stream);
/* adjoint: dX [R,K] = dZ W[:, :K] (NULL: skipped); dW [N, K + 1] = [ dZ^T X | dZ^T xs ]; db [N] (NULL: no bias); xs has no gradient */
int pdp_train_linear_s_backward(const float *dY, const float *Y, const float *X, const float *xs, int64_t R, int K, int64_t ldx, const float *W, int N, int act,
                                float *dZ, float *dX, int64_t lddx, float *dW, float *db, void *stream);
/* replaces: torch.mm(mask, state) of MessageAggregator.forward (util.py:60): x [E,A] -> out [rows,A], ordered sum over the edges of every
 * variable (by_variable != 0) or clause */
int pdp_train_row_sum(pdp_problem *p, int by_variable, const float *x, int A, float *out, void *stream);
/* replaces: torch.mm(mask_transpose, aggregated) - state (util.py:63-69): rows [rows,A] -> out [E,A] = rows[row(e)] - x[e] (x NULL: gather only).
 * The adjoint of the exclude-self aggregation is the same pair of calls on the gradient; of the plain row sum, the gather. */
int pdp_train_row_spread(pdp_problem *p, int by_variable, const float *rows, const float *x, int A, float *out, void *stream);
/* replaces: nn.GRUCell forward (pdp_decimate.py:38-41,70-83): x [R,Kx], h [R,H], W_ih [3H,Kx], W_hh [3H,H] -> hnew [R,H];
 * saved [R,4H] = r | z | n | W_hn h + b_hn for the adjoint; scratch [R,6H] */
int pdp_train_gru(const float *x, const float *h, const float *W_ih, const float *W_hh, const float *b_ih, const float *b_hh, int64_t R, int Kx, int H,
                  float *hnew, float *saved, float *scratch, void *stream);
/* the same forward for the hidden-128 cells of the shipped models (x = [state [R,dx] | sign [R]], dx = 128: np-nd-np, dx = 2 or 3: p-nd-np) in ONE
 * launch of the pipelined inference kernel, which also writes
 * saved [R,4H]: no gi / gh round trip through memory.  Weights in the descriptor layout of the inference cell -- transposed, padded; R a multiple of 64 (the caller runs
 * pdp_train_gru on the rows behind the last full tile). */
int pdp_train_gru_fused(const pdp_gru_desc *d, const float *state, const float *sign, const float *h, int64_t R, float *hnew, float *saved, void *stream);
/* adjoint: dhnew [R,H] -> dx [R,Kx], dh [R,H], dW_ih, dW_hh, db_ih [3H], db_hh [3H]; scratch [R,6H] (a [R,7H] block is fine) */
int pdp_train_gru_backward(const float *dhnew, const float *saved, const float *x, const float *h, const float *W_ih, const float *W_hh, int64_t R, int Kx,
                           int H, float *dx, float *dh, float *dW_ih, float *dW_hh, float *db_ih, float *db_hh, float *scratch, void *stream);
/* the adjoint for a cell whose input is [state [R,Ks] | xs [R]] held apart (W_ih [3H, Ks + 1]): dstate [R,Ks], dW_ih [3H, Ks + 1]; scratch [R,6H] */
int pdp_train_gru_backward_s(const float *dhnew, const float *saved, const float *state, const float *xs, const float *h, const float *W_ih, const float *W_hh,
                             int64_t R, int Ks, int H, float *dstate, float *dh, float *dW_ih, float *dW_hh, float *db_ih, float *db_hh, float *scratch, void *stream);
/* adjoint of pdp_sp_propagate_adapted (the adaptor form of SurveyPropagator.forward, pdp_propagate.py:163-221, as the training path runs it: no
 * active mask): upstream g_q [E,3] (surveys) and g_eta [E] (column 0 of the function state; the force column carries no gradient, torch.sign) ->
 * d_xlog [E] (gradient of the log-domain clause message = logsigmoid of the function projector's output) and d_eta_in [E] (gradient of
 * fs2[:,0] = sigmoid of the variable projector's first output).  xlog / fs2 / edge_mask as handed to the forward. */
int pdp_train_sp_adapted_backward(pdp_problem *p, const float *xlog, const float *fs2, const float *edge_mask, float pi, const float *g_q,
                                  const float *g_eta, float *d_xlog, float *d_eta_in, void *stream);
/* adjoint of pdp_sat_loss (SatLossEvaluator.forward, util.py:178-197) with respect to the prediction: dpred [V] = upstream * d loss / d pred */
int pdp_sat_loss_grad(pdp_problem *p, const float *pred, float coeff, float eps, int sharpness, float upstream, float *dpred, void *stream);

/* ---- the reference's L0 primitives in their original call shape (a sparse COO mask instead of a problem handle) ---------------------
 * For plug-ins written against the reference's API: they hand util.sparse_max / sparse_argmax /
 * sparse_smooth_max and MessageAggregator.forward the torch sparse masks of SATProblem (or masks they built themselves).  The host maps
 * the masks SATProblem built back to the resident layout (the entry points above); any other mask arrives here as its index / value
 * arrays.  No pdp_problem is involved.
 * replaces: util.sparse_max / sparse_argmax (util.py:257-275) for an arbitrary mask [n_rows, n_cols] given as nnz (row, col) pairs in
 * the order of mask._indices(); x [nnz] is paired with the entries in that order (the reference builds
 * sparse(mask._indices(), x - x.min() + 1).to_dense()); out [n_cols]; an empty column yields 0 (arg-max) / x.min() - 1 (max); NaN counts
 * as the largest value, ties go to the smallest row.  scratch: uint64 [n_cols + 2]. */
int pdp_coo_max(const int64_t *rows, const int64_t *cols, int64_t nnz, const float *x, int64_t n_rows, int64_t n_cols, uint64_t *scratch,
                float *out, void *stream);
int pdp_coo_argmax(const int64_t *rows, const int64_t *cols, int64_t nnz, const float *x, int64_t n_rows, int64_t n_cols, uint64_t *scratch,
                   int64_t *out, void *stream);
/* row offsets [n_rows + 1] of entries sorted by row (a coalesced torch sparse tensor) */
int pdp_coo_row_ptr(const int64_t *sorted_rows, int64_t nnz, int64_t n_rows, int64_t *row_ptr, void *stream);
/* replaces: torch.mm(mask, X) of util.py:60,63 for an arbitrary mask given as row offsets + columns (+ values, NULL: ones) of its row-sorted
 * entries: out [n_rows, d] = mask X [., d] (row stride ldx) - sub (NULL: nothing subtracted; the exclude-self form of util.py:63-69);
 * a row's entries are added in ascending entry order */
int pdp_csr_matmul(const int64_t *row_ptr, const int64_t *cols, const float *vals, int64_t n_rows, const float *X, int d, int64_t ldx,
                   const float *sub, float *out, void *stream);
/* replaces: util.sparse_smooth_max (util.py:282-286) for an arbitrary mask and alpha: x [n_cols] -> out [n_rows] */
int pdp_csr_smooth_max(const int64_t *row_ptr, const int64_t *cols, const float *vals, int64_t n_rows, const float *x, float alpha,
                       float *out, void *stream);

/* ---- kernel timing (measurement only: bench.py's per-kernel roofline lines) ------------------------------
 * When enabled, the library brackets the kernels named below with HIP events ON THEIR LAUNCH STREAM (the stream handed to the entry
 * point); pdp_kernel_timing_read synchronises on the recorded events and returns, per key, the summed device time in ms and the number
 * of launches since the last read.  No effect on results; off by default. */
enum { PDP_TK_AGG_PRE = 0,      /* k_agg_pre_wave / k_agg_pre_res / k_agg_pre: first half of a MessageAggregator (two layers per edge) */
       PDP_TK_ROW_SUM = 1,      /* k_row_sum: per-row sum of the pre-transformed edges */
       PDP_TK_AGG_POST = 2,     /* k_agg_post_pf / k_agg_post_wave (hidden 150) / k_agg_post: second half (two layers per edge, masked blend) */
       PDP_TK_GRU = 3,          /* k_gru_pipe (hidden 128) / k_gru_wave (hidden 150) / k_gru */
       PDP_TK_PREDICT_HEAD = 4, /* k_predict_rows: per-variable layers + perceptron head of NeuralPredictor */
       PDP_TK_WALKSAT = 5,      /* k_walksat<uint16_t, 256>: the persistent LDS-resident Walk-SAT launch of pdp_local_search */
       PDP_TK_SP_ADAPTORS = 6,  /* k_sp_adaptors: the two projections in front of the adaptor form of the SP sweep (p-nd-np) */
       PDP_TK_SP_SWEEP = 7,     /* k_sp_propagate<false|true>: the step-wise SP sweep (pdp_sp_propagate / pdp_sp_propagate_adapted) */
       PDP_TK_COUNT = 8,
       /* name-only keys (their times come back through pdp_solve_args) */
       PDP_KN_SP_SOLVE = 8,     /* pass 1 of pdp_sp_solve (k_sp_solve_lds<...> / k_sp_solve<...>) */
       PDP_KN_SP_REPLAY = 9,    /* its NaN-poison replay instantiation */
       PDP_KN_COUNT = 10 };
int pdp_kernel_timing(int enable);
/* the name (with template arguments, as rocprofv3 prints it) of the kernel the library launched last for `key`; empty before the first launch */
int pdp_kernel_name(int key, char *buf, int len);
int pdp_kernel_timing_read(float *ms_host /*[PDP_TK_COUNT]*/, int32_t *launches_host /*[PDP_TK_COUNT]*/);

/* ---- math probes (tests: device exp/log must equal the host header bit for bit) ---------------------- */
int pdp_math_apply(int fn, const float *x, float *y, int64_t n, void *stream);

#ifdef __cplusplus
}
#endif
#endif /* PDP_HIP_H */
