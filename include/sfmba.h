/*
 * sfmba.h -- C-ABI of libsfmba.so, the MI355X (gfx950) bundle-adjustment back end.
 *
 * Drop-in boundary: the one call the reference makes into its optimiser,
 *
 *     result = least_squares(compute_residuals, x0, jac_sparsity=..., verbose=..., x_scale='jac',
 *                            ftol=tol, method='trf',
 *                            args=(n_cam, n_points, camera_indices, pt_indices, pt2ds, K))
 *                                                     -- /root/reference/sfm_lite/sfm.py:266-268
 *
 * plus the model function it hands over, compute_residuals
 *                                                     -- /root/reference/sfm_lite/bundle_adjustment.py:35-42
 *
 * The reference is pure Python, so the "FFI" a maintainer adds is a ctypes binding of exactly these
 * entry points (shown in INTEGRATION.md; shipped as sfm-python_amd/sfmba/_capi.py).  Plain C types
 * only: caller-owned, C-contiguous HOST arrays are borrowed for the duration of a call and copied to
 * HBM inside it; nothing returned is library-owned except the handle and the error string.
 *
 * Return codes: 0 OK; -1 bad argument / shape / index out of range; -2 residuals not finite at x0
 * (scipy raises ValueError there, SCIPY/optimize/_lsq/least_squares.py:844-845); -3 HIP failure;
 * -4 out of memory; -5 collective failure (RCCL, the all-reduce callback, peer mapping, or a direct
 * all-reduce that gave up waiting for a peer).  The solver outcome is NOT an error:
 * it is sfmba_result.status, scipy's 0..4 (SCIPY/optimize/_lsq/least_squares.py:18-25).
 *
 * Threading: one handle is not thread-safe; distinct handles are independent; every entry point
 * selects the handle's device on entry, so it may be called from any thread (the reference's GUI runs
 * BA on a worker thread, /root/reference/app.py:80-85).
 */
#ifndef SFMBA_H
#define SFMBA_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct sfmba_handle sfmba_handle;

/* Options of sfmba_solve.  Fill with sfmba_default_options() first.
 * ftol/xtol/gtol/max_nfev/verbose mirror the least_squares kwargs of sfm.py:266-268 (xtol, gtol,
 * max_nfev at scipy's defaults 1e-8, 1e-8, 100*n when the caller passes none). */
typedef struct sfmba_options {
    double  ftol;          /* scipy ftol; reference passes ba_tol (1e-10)                         */
    double  xtol;          /* 1e-8                                                               */
    double  gtol;          /* 1e-8                                                               */
    int64_t max_nfev;      /* <=0: 100 * n  (SCIPY trf.py:437-438)                               */
    int32_t verbose;       /* 0 silent, 1 summary (printed by the host shim), 2 iteration table  */
    int32_t max_iter;      /* <=0: unlimited; otherwise stop after this many outer iterations    */
    double  pcg_tol;       /* forcing term of the inexact step: the Schur PCG stops when the
                              preconditioned residual norm has dropped by this factor (default 1e-2;
                              a tenth of it when 6 n_cameras <= 128 and the iterations run inside one
                              workgroup; DESIGN.md section 3 compares this with scipy's LSMR)        */
    int32_t pcg_max_iter;  /* <=0: 2 * 6 * n_cameras                                             */
    int32_t pcg_check_every; /* host polls the device-side convergence flag every k iterations   */
    double  reg_min;       /* floor of the Levenberg-Marquardt term (1e-6), see DESIGN.md         */
    int32_t profile;       /* 1: bracket every residual+Jacobian launch with HIP events          */
    int32_t reserved;
    double  pcg_tol_max;   /* > pcg_tol: the forcing term adapts to the progress of the outer iteration,
                              eta_k = min(pcg_tol_max, max(pcg_tol, |g_h(x_k)| / |g_h(x_k-1)|)), eta_0 =
                              pcg_tol_max (default 0.1; Eisenstat-Walker: loose while the scaled gradient
                              still drops slowly, pcg_tol once it drops fast).  <= pcg_tol: fixed term.
                              Not applied to the in-workgroup iterations of the few-camera path.    */
} sfmba_options;

typedef struct sfmba_result {
    double  cost;          /* 0.5 |f|^2 at the returned x                                        */
    double  cost0;         /* 0.5 |f|^2 at x0                                                    */
    double  optimality;    /* |J^T f|_inf                                                        */
    double  rmse;          /* sqrt(2 cost / (2 N_total))                                         */
    double  rmse0;
    int64_t nfev;
    int64_t njev;
    int64_t iterations;    /* outer (trust-region) iterations                                    */
    int64_t pcg_iterations;/* total inner PCG iterations                                         */
    int32_t status;        /* 0 max_nfev/max_iter, 1 gtol, 2 ftol, 3 xtol, 4 ftol+xtol           */
    int32_t reserved;      /* number of PCG solves that stopped on a numerical breakdown         */
    double  seconds_total; /* wall time of the call, host clock, H2D/D2H included                */
    double  seconds_device;/* wall time between upload and download                              */
    double  resjac_avg_us; /* HIP-event average of the residual+Jacobian kernel (profile=1)      */
    int64_t resjac_launches;
    double  last_step_norm;
    double  last_reg;
} sfmba_result;

/* All-reduce hook for observation-sharded problems (one process per GPU).  `dev_ptr` points into the
 * exchange arena registered with sfmba_set_exchange; `count` doubles are reduced in place over all
 * ranks on the handle's stream; op 0 = sum, 1 = max.  Return 0 on success. */
typedef int (*sfmba_allreduce_fn)(void* ctx, void* dev_ptr, int64_t count, int32_t op);

/* Sink for the lines of the verbose = 2 iteration table (scipy prints it through Python's sys.stdout,
 * SCIPY/optimize/_lsq/common.py:545-563; the host shim registers a callback that does the same, so that
 * redirect_stdout / notebook capture see the table in order with the summary).  `line` has no trailing newline and
 * is valid during the call only.  fn = NULL: the C library's stdout. */
typedef void (*sfmba_print_fn)(void* ctx, const char* line);

/* ---- lifetime ---------------------------------------------------------------------------- */
int  sfmba_create(sfmba_handle** out, int device_id);
void sfmba_destroy(sfmba_handle* h);
const char* sfmba_last_error(const sfmba_handle* h);     /* valid until the next call on h */
void sfmba_default_options(sfmba_options* opt);
int  sfmba_set_print(sfmba_handle* h, sfmba_print_fn fn, void* ctx);
/* Run every kernel of `h` on this hipStream_t (default: a stream the handle owns). */
int  sfmba_set_stream(sfmba_handle* h, void* hip_stream);

/* Storage precision of the per-observation streams (uv, r, Jacobian): 64 (default) or 32.  Arithmetic
 * and every accumulation are fp64 in both modes; 32 halves the bytes of the sweeps (BASELINE.json
 * config 5).  Takes effect at the next sfmba_set_problem. */
int  sfmba_set_precision(sfmba_handle* h, int32_t storage_bits);

/* ---- problem (the args= tuple of sfm.py:268) ---------------------------------------------- */
/* camera_indices, point_indices: (N) int64; points_2d: (N,2) float64 (caller converts the
 * reference's int64 pixels, bundle_adjustment.py:41 promotes them the same way); K: 3x3 row-major.
 * Indices are range-checked here (the reference's fancy indexing would raise IndexError,
 * bundle_adjustment.py:40).  Any observation order is accepted; point-major (what
 * Graph.pt3ds_pt2ds produces, graph.py:186-191) is the fast path.
 * A new problem returns the handle to single-process operation: a registered exchange callback, RCCL
 * communicator or direct link is dropped and has to be set up again after this call.  The drop is not silent: when
 * a transport was active, every compute call on the handle fails with -1 until one is set up again
 * (sfmba_set_exchange / sfmba_comm_init / sfmba_p2p_attach) or single-rank use of the shard is acknowledged
 * (sfmba_set_exchange(h, NULL, 0, NULL, NULL, 0), sfmba_comm_destroy, sfmba_p2p_detach). */
int  sfmba_set_problem(sfmba_handle* h, int64_t n_cameras, int64_t n_points, int64_t n_obs,
                       const int64_t* camera_indices, const int64_t* point_indices,
                       const double* points_2d, const double* K);
/* Incremental re-use: the reference calls BA once per fused edge on a growing reconstruction
 * (/root/reference/sfm_lite/sfm.py:59-71); once all cameras are registered a new edge only appends points and
 * observations, so the arrays of one call start with those of the previous call.  Every sfmba_set_problem compares the
 * arrays with the previous problem of the handle (kept converted in pinned memory and in HBM), finds the first
 * observation that differs and uploads from there on only; structure tables are always rebuilt from the complete
 * arrays, so results are bitwise those of a fresh handle.  sfmba_problem_reuse reports what the last call re-used. */
int  sfmba_problem_reuse(const sfmba_handle* h, int64_t* obs_reused, int64_t* obs_uploaded);
/* Same with the pixels as the reference holds them, (N,2) int64 (graph.py:112-113): saves the caller a
 * converted copy. */
int  sfmba_set_problem_i64(sfmba_handle* h, int64_t n_cameras, int64_t n_points, int64_t n_obs,
                           const int64_t* camera_indices, const int64_t* point_indices,
                           const int64_t* points_2d, const double* K);

/* Cameras held still: the `fixed_camera_indices` of create_sparsity_matrix (bundle_adjustment.py:6,13-14), whose six
 * Jacobian columns the pattern leaves empty -- scipy's finite differences then never fill them, so the gradient, the
 * step and hence the parameters of those cameras stay as they are, and x_scale='jac' gives them scale 1
 * (SCIPY/optimize/_lsq/common.py:598-610).  The list is read at the NEXT sfmba_set_problem and stays in force until
 * it is replaced (n_fixed = 0 clears it); indices are range-checked there.  Their observations still count in the
 * residual, the cost and the point blocks. */
int  sfmba_set_fixed_cameras(sfmba_handle* h, const int64_t* camera_indices, int64_t n_fixed);

/* Observation sharding: this handle holds the local shard (its own points + their observations,
 * all cameras replicated); n_obs_total counts all shards.  `arena` is device memory of at least
 * sfmba_exchange_doubles(n_cameras) doubles that `fn` can all-reduce (e.g. a torch tensor).
 * Pass fn = NULL to return to single-process operation. */
int64_t sfmba_exchange_doubles(int64_t n_cameras);
int  sfmba_set_exchange(sfmba_handle* h, void* arena, int64_t arena_doubles,
                        sfmba_allreduce_fn fn, void* ctx, int64_t n_obs_total);

/* Native collective path: the library all-reduces its exchange arena itself with RCCL over xGMI
 * (librccl.so.1 is opened at run time; ncclAllReduce is enqueued on the handle's stream, so there is
 * no host synchronisation and no interpreter on the data path).  Rank 0 obtains the 128-byte id,
 * the host shim distributes it to all ranks (torch.distributed broadcast in sfmba.dist), every rank
 * calls sfmba_comm_init.  world == 1 is valid.  sfmba_comm_destroy returns to single-process mode. */
int  sfmba_comm_get_unique_id(void* id128_out);
int  sfmba_comm_init(sfmba_handle* h, const void* id128, int32_t rank, int32_t world,
                     int64_t n_obs_total);
int  sfmba_comm_destroy(sfmba_handle* h);

/* Direct all-reduce over peer-mapped device memory (xGMI inside a node), used in preference to RCCL or
 * the callback for every exchanged vector once attached: the solver reduces a dozen small vectors (a few
 * scalars to 27*n_cameras doubles) per iteration, which is latency, not bandwidth.  Collective set-up,
 * after sfmba_set_problem and sfmba_comm_init / sfmba_set_exchange on every rank:
 *   sfmba_p2p_export  allocates this rank's staging buffer and returns its 64-byte hipIpcMemHandle_t;
 *   (the caller all-gathers the handles over any channel, rank order)
 *   sfmba_p2p_attach  maps the peers' buffers and runs a 24-round self-test (the three message sizes of
 *                     the solver, chains of back-to-back collectives); returns -5 and detaches when
 *                     mapping or the self-test fails (the previous transport stays).  Its last collective settles
 *                     what the ranks must agree on for the per-camera sums to be exchanged INSIDE the kernels that
 *                     form them (pass B of the Schur product, the camera blocks, the reduced right-hand side): that
 *                     no rank's shard has a camera cut into several chunks, and how many ranks sit on one physical
 *                     GPU (a rehearsal: waiting workgroups of one rank must not fill the card the other needs).
 *                     Otherwise those sums keep collective launches of their own -- on every rank alike.
 * A wait gives up after 30 s (60 s for the first collective of a solve, the rendezvous) and fails the solve with -5;
 * the text of sfmba_last_error names the exchange and the camera whose peer value did not arrive.
 * All ranks must attach or none: agree on the minimum of the return codes and call sfmba_p2p_detach on
 * every rank if any failed.  world <= 16.  Results are summed in rank order: bitwise equal on all ranks. */
int  sfmba_p2p_export(sfmba_handle* h, int32_t world, void* handle64_out);
int  sfmba_p2p_attach(sfmba_handle* h, const void* handles_world_x_64, int32_t rank, int32_t world);
int  sfmba_p2p_detach(sfmba_handle* h);
int64_t sfmba_p2p_calls(const sfmba_handle* h);      /* collectives served by the direct path so far */
/* Running totals since sfmba_create: kernel launches enqueued by the library and collectives performed (any
 * transport).  Differences around a solve give launches / collectives per outer iteration (bench.py). */
int  sfmba_get_counters(const sfmba_handle* h, int64_t* kernel_launches, int64_t* collectives);
/* PCG iterations of every outer iteration of the last completed sfmba_solve on this handle (the record the next
 * solve's speculative launches are sized from); returns the number of outer iterations, writes min(that, cap)
 * entries.  What the full-size parity tests compare with the recorded oracle runs (tests/golden/oracle_cfg*.json). */
int32_t sfmba_get_pcg_history(const sfmba_handle* h, int32_t* out, int32_t cap);

/* ---- compute_residuals (bundle_adjustment.py:35-42) ----------------------------------------- */
/* x: (6C+3P) float64 -> r_out: (2N) float64, interleaved x,y in the caller's observation order. */
int  sfmba_residuals(sfmba_handle* h, const double* x, double* r_out);

/* Residual and analytic Jacobian blocks (replaces scipy's sparse 2-point finite differences,
 * SCIPY/optimize/_numdiff.py:628-705).  Jc: (N,2,6) d r/d(rotvec,T); Jp: (N,2,3) d r/d X. */
int  sfmba_residual_jacobian(sfmba_handle* h, const double* x, double* r_out, double* Jc_out,
                             double* Jp_out);

/* ---- least_squares(method='trf', x_scale='jac') (sfm.py:266-268) ---------------------------- */
/* x_inout: x0 on entry, result.x on success (untouched on failure). */
int  sfmba_solve(sfmba_handle* h, double* x_inout, const sfmba_options* opt, sfmba_result* out);
/* The same with the start and the result in separate arrays (x_out may be x0): a caller that keeps x0 -- scipy's
   least_squares does not modify its argument -- saves the copy it would otherwise make for the in/out form. */
int  sfmba_solve_from(sfmba_handle* h, const double* x0, double* x_out, const sfmba_options* opt, sfmba_result* out);
/* After a successful sfmba_solve: result.fun (2N) and result.grad (6C+3P); either may be NULL. */
int  sfmba_get_fun_grad(sfmba_handle* h, double* fun_out, double* grad_out);

/* ---- measurement / test entry points -------------------------------------------------------- */
/* `reps` back-to-back launches of one kernel at x, bracketed by HIP events on the handle's stream.
 * which: 0 residual+Jacobian sweep (with the point blocks V_p, g_p it leaves behind), 1 residual-only sweep,
 *        2 the camera pass of the normal equations (U_c, g_c, and the point rows the sweep's tiles cut),
 *        3 one implicit Schur product (pass A + pass B), 4 pass A alone, 5 pass B alone, 6 the reduced
 *        right-hand-side pass, 7 the residual+Jacobian sweep with its point-block sums switched off,
 *        10 a streaming-store fill of the Jacobian buffer (ceiling probe).
 * avg_us: average duration of one repetition. */
int  sfmba_time_kernel(sfmba_handle* h, const double* x, int32_t which, int32_t reps, double* avg_us);
/* Normal-equation blocks at x: U (C,21 upper triangle row-major), V (P,6 upper), gc (C,6), gp (P,3). */
int  sfmba_normal_blocks(sfmba_handle* h, const double* x, double* U, double* V, double* gc,
                         double* gp);
/* y = S v with S = U + diag(dc) - W (V + diag(dp))^-1 W^T at x (v, dc, y: 6C; dp: 3P). */
int  sfmba_schur_matvec(sfmba_handle* h, const double* x, const double* dc, const double* dp,
                        const double* v, double* y);

/* Few-camera path (6 n_cameras <= 128, the reference's own problem sizes): at x, with the diagonals dc (6C) and dp
 * (3P), form S = U + diag(dc) - W (V + diag(dp))^-1 W^T (S_out: (6C)^2 row-major, may be NULL) and solve S y = rhs
 * with the in-LDS PCG run to the end (relative tolerance 1e-14; sol_out: 6C). */
int  sfmba_dense_schur(sfmba_handle* h, const double* x, const double* dc, const double* dp, const double* rhs,
                       double* S_out, double* sol_out);

/* Host-only helper for the CPU test-suite (no GPU needed): the 2-D trust-region subproblem of
 * SCIPY/optimize/_lsq/common.py:171-219.  B3 = (B00, B01, B11), g2, Delta -> p2; returns 1 when the Newton step
 * lies inside the region, 0 for a boundary solution. */
int  sfmba_tr2d_solve(const double* B3, const double* g2, double Delta, double* p2);
/* Test / diagnostic hooks (nothing in the library reads the environment).  -1 (or 0 where noted) = the library's own
 * choice.  Placement options (P) take effect at the next sfmba_set_problem, the others at the next solve.
 *   "dense"          P  0: implicit Schur product + launched PCG although the reduced camera matrix would be formed
 *                       and solved inside one workgroup (6 n_cameras <= 128)
 *   "sweep_rc"       P  0: pass A of the Schur product reads the stored Jacobian instead of recomputing the blocks from
 *                       the camera table; 2: recomputes them from a table in GLOBAL memory (the form of > 1100 cameras)
 *                       whatever the camera count
 *   "tab_lds", "vec_lds" P  0: camera table / camera vector read from L2 although they would fit the LDS
 *   "pcg_fused"      P  0: the PCG update as a kernel of its own instead of the prologue of pass A
 *   "pcg_local"         0: the fused PCG keeps its whole update in pass A's prologue (no per-camera tail in pass B)
 *   "pcg_split"         1: the per-camera tail of the local form in a kernel of its own (k_pcg_tail) on one rank too;
 *                       0: sharded / multi-chunk solves keep the general prologue or k_pcg_update
 *   "precond"           0: block-Jacobi preconditioner from U + D instead of the Schur-diagonal blocks
 *   "cam_chunk"      P  > 0: chunk length of the camera-major kernels (forces multi-chunk cameras + k_cam_combine)
 *   "xcd_chunks"     P  1 / 0: pass B's camera lists cut at the eight point-range boundaries (one piece per XCD) whatever
 *                       the problem size (default: from 250k points on)
 *   "rhsrec"         P  1 / 0: the rhs + preconditioner pass gathers dedicated 128-byte point records whatever the size
 *   "cost_rider"        0: the trial cost is summed and posted by a k_finish launch of its own instead of riding with
 *                       the normal-block launch
 *   "pcg_mixed"      P  1 / 0: fp32 operands with fp64 accumulation in the implicit Schur product (default: with fp32
 *                       storage, sfmba_set_precision); "pcg_mixed_b" 0: pass B keeps its fp64 point records
 *   "pcg_inline"        0: sharded solves over the direct link keep the all-reduces of the per-camera sums as launches
 *                       of their own (default: exchanged by the producing workgroups; ranks that SHARE one device --
 *                       rehearsals -- get 0 by themselves once their camera workgroups together exceed the device's
 *                       resident slots: settled at sfmba_p2p_attach)
 *   "xcd_cam"        P  0: K3 and the rhs + preconditioner pass keep one workgroup per camera where pass B takes the
 *                       XCD-aware chunk table ("xcd_chunks"; default from 250k points on: all three one wave per chunk)
 *   "pcg_skip_last"     0: the launches that a replayed record says find the PCG solve finished are enqueued all the
 *                       same; 2: only the pass B is left out (default: neither the last pass A -- k_backsub's prologue
 *                       does its update -- nor the pass B behind it is enqueued; both are owed if the record turns out
 *                       too short)
 *   "cm_device"      P  1 / 0: camera-major order sorted on the device / on the host (default: device from 64k
 *                       observations on); "packed_upload" P 1 / 0: observation arrays uploaded packed (same default)
 *   "jfree"          P  1: the J-free iteration (measurement): K1 does not write the Jacobian, k_jdot / k_backsub
 *                       recompute its blocks (needs the camera table in LDS)
 *   "pcg_guess_bias"    added to the number of speculatively enqueued PCG iterations (negative: force misses)
 *   "wait_deadline_s"   a hand-off not posted within this many seconds fails the solve with -3 (default 120)
 *   "p2p_delay_ms"      sleep before the first collective of a solve (late-peer test)
 *   "p2p_timeout_ms"    > 0: overrides both time-outs of the direct all-reduce (test)
 *   "trace_pcg", "trace_stalls", "trace_timing"   stderr diagnostics */
int  sfmba_debug_option(sfmba_handle* h, const char* name, int64_t value);

#ifdef __cplusplus
}
#endif
#endif /* SFMBA_H */
