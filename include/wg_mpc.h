/*
 * wg_mpc.h -- C ABI of the MI355X-native ZMP-MPC hot path (libwg_mpc.so).
 *
 * Drop-in boundary for the per-tick inner loop of jrl-walkgen's Herdt-2010
 * pattern generator.  Plain pointers and sizes only; no C++/torch types.
 * Every entry point names the reference interface it replaces.
 *
 * Conventions
 *   - return value: 0 on success, negative on error (wg_last_error() has text);
 *   - per-QP `ifail` keeps the QL codes of the reference
 *     (src/privatepgtypes.hh:349-356): 0 ok, 1 too many iterations,
 *     2 accuracy insufficient, 5 workspace too short, >10 inconsistent
 *     constraints (10 + constraint code);
 *   - all matrices column-major (Fortran), one QP after the other, strides
 *     given by (nmax, mmax) exactly as ql0001_ takes them;
 *   - `_dev` variants take DEVICE pointers and a hipStream_t (passed as void*)
 *     and are asynchronous; the others take HOST pointers and are synchronous;
 *   - the library never falls back to a CPU path: without a usable HIP device
 *     every compute entry point fails with WG_ERR_NO_DEVICE.
 */
#ifndef WG_MPC_H
#define WG_MPC_H

#include <stddef.h>

#ifdef __cplusplus
extern "C" {
#endif

#define WG_OK 0
#define WG_ERR_NO_DEVICE (-1)
#define WG_ERR_BAD_ARG (-2)
#define WG_ERR_HIP (-3)
#define WG_ERR_TOO_LARGE (-4)
#define WG_ERR_BUSY (-5)      /* WG_OVERLAP_STRICT=1 only: a launch of the same context is still in flight on another stream */

/* Library / device management ------------------------------------------- */

/* Select the HIP device this process drives (one process per GPU). */
int wg_init(int device_ordinal);
void wg_shutdown(void);
const char *wg_last_error(void);
/* ABI version, bumped on any signature change. */
int wg_abi_version(void);
/* LDS bytes one QP of size (n, m) occupies in the solver kernel; lets callers
 * reason about occupancy (160 KiB per CU on gfx950). */
size_t wg_qp_lds_bytes(int n, int m);

/* Sharding (host arithmetic) ---------------------------------------------------------------------------------------------
 *
 * Gaits are independent (the reference has no coupling between pattern generators), so a fleet of `total` gaits is split
 * over `world` ranks -- one process per GPU -- by contiguous index range, the remainder spread over the first ranks:
 * rank r owns the global gaits [*lo, *hi).  Every rank derives per-gait inputs (seeds, references) from the GLOBAL index,
 * so the job's results do not depend on how it was split.  The only thing ranks exchange is the constant block
 * wg_model_t, which rank 0 broadcasts once (host/fleet_bench.cpp: ncclBroadcast; jrl-walkgen_amd/shard.py: the same
 * through torch.distributed). */
int wg_shard_range(long long total, int rank, int world, long long *lo, long long *hi);

/* Contexts ---------------------------------------------------------------------------------------------------------------
 *
 * Everything the library keeps between calls lives in a context: the configured models with the device copies of their
 * tables (wg_mpc_configure, wg_pldp_configure, wg_dimitrov_configure, wg_preview_configure), the per-launch workspaces of
 * the tick kernels (work queue of the multi-tick launch, per-block solver slots) and the staging buffers of the
 * host-pointer entry points.  The reference keeps the same things per object (one ZMPVelocityReferencedQP, one
 * PreviewControl ... per PatternGeneratorInterface); a context is what one such object owns here.
 *
 *   - every entry point NAME(args) below has a form NAME_ctx(ctx, args) with identical semantics on that context;
 *     NAME(args) itself works on a process-wide default context (device 0 unless wg_init chose another);
 *   - contexts are independent: two contexts may hold different models (N = 16 and N = 32, two robots, two preview
 *     windows) and their launches may overlap on different streams;
 *   - launches on ONE context share its workspaces: keep them on one stream (or order them with events).  Host-pointer
 *     entry points are synchronous and serialised per context: each context owns one non-blocking stream on which they copy
 *     in, launch and copy out, and they wait for THAT stream only -- never for the device: a host call on one context does
 *     not stall, and is not stalled by, the launches of another context or of the caller's own streams (two
 *     PatternGeneratorInterface objects, or a facade object beside a fleet, run side by side).  The same holds for the
 *     *_configure entry points: re-configuring waits for this context's earlier launches (their events), nobody else's;
 *   - a context belongs to one device; its entry points make that device current for the calling thread;
 *   - wg_ctx_destroy waits for the device, then frees everything the context owns. */
typedef struct wg_ctx wg_ctx_t;
int wg_ctx_create(int device_ordinal, wg_ctx_t **ctx);
void wg_ctx_destroy(wg_ctx_t *ctx);
int wg_ctx_device(const wg_ctx_t *ctx);
/* Launches of one context that arrive on different streams are ordered by the library (see wg_mpc_tick_batch_dev).  `on` != 0
 * makes the context refuse them instead (WG_ERR_BUSY); the initial value is WG_OVERLAP_STRICT of the environment, read once when
 * the context is created.  wg_overlap_serialised: how many launches of this context were ordered behind a launch of another
 * stream so far -- a caller that expected two streams to overlap reads its lost concurrency here (WG_OVERLAP_NOTE=1 in the
 * environment also prints one line to stderr at the first occurrence). */
int wg_set_overlap_strict(int on);
long long wg_overlap_serialised(void);

/* Batched dense QP solve -------------------------------------------------
 *
 * Replaces, batched over B independent problems,
 *     ql0001_(m, me, mmax, n, nmax, mnn, c, d, a, b, xl, xu, x, u, iout,
 *             ifail, iprint, war, lwar, iwar, liwar, eps1)
 *     src/Mathematics/qld.hh:27-31, called from QPProblem::solve
 *     src/ZMPRefTrajectoryGeneration/qp-problem.cpp:275-279
 * with iwar[0] = 1 (solver factorises the Hessian itself):
 *
 *     minimise  1/2 x'Cx + d'x   s.t.  A_j x + b_j  = 0  (j <  me)
 *                                      A_j x + b_j >= 0  (me <= j < m)
 *                                      xl <= x <= xu
 *
 *   B        number of QPs
 *   nmax     leading dimension of every C, stride of d/xl/xu/x/iact (>= max n)
 *   mmax     leading dimension of every A, stride of b             (>= max m)
 *   n,m,me   per-QP sizes, B ints each (n may be NULL => all nmax;
 *            m NULL => all mmax-1, the reference's convention m = mmax-1;
 *            me NULL => all 0)
 *   C        B * nmax*nmax     d, xl, xu   B * nmax
 *   A        B * mmax*nmax     b           B * mmax
 *   eps      the reference passes 1e-8 (qp-problem.cpp:260)
 * outputs
 *   x        B * nmax
 *   u        B * (mmax + 2*nmax): multipliers [general m | lower n | upper n]
 *            packed with the QP's own m and n like ql0001_'s u (may be NULL)
 *   ifail    B
 *   n_iter   B   ql0002's iteration counter                 (may be NULL)
 *   iact     B * nmax final active set in activation order, QL codes
 *            (1..m row, m+1..m+n lower bound, m+n+1.. upper), 0 padded
 *            -- what ql0001_ leaves in iwar[0..nact)          (may be NULL)
 *   nact     B                                               (may be NULL)
 *   hist     B * hist_cap add(+code)/drop(-code) log          (may be NULL)
 *   hist_len B   number of events (can exceed hist_cap)       (NULL iff hist NULL)
 * With more QPs than the device keeps resident, a batch that follows another one of the same size on the same C array starts
 * its QPs in the order of decreasing iteration count of that previous batch (an MPC loop: problem k of consecutive calls is
 * the same robot a tick later).  Scheduling only: no result depends on it; WG_QL_LPT=0 keeps index order. */
int wg_qp_solve_batch(int B, int nmax, int mmax, const int *n, const int *m,
                      const int *me, const double *C, const double *d,
                      const double *A, const double *b, const double *xl,
                      const double *xu, double eps, double *x, double *u,
                      int *ifail, int *n_iter, int *iact, int *nact, int *hist,
                      int hist_cap, int *hist_len);

int wg_qp_solve_batch_dev(int B, int nmax, int mmax, const int *n, const int *m,
                          const int *me, const double *C, const double *d,
                          const double *A, const double *b, const double *xl,
                          const double *xu, double eps, double *x, double *u,
                          int *ifail, int *n_iter, int *iact, int *nact,
                          int *hist, int hist_cap, int *hist_len,
                          void *hip_stream);


/* ------------------------------------------------------------------------
 * Herdt-2010 MPC tick, batched over independent gaits
 *
 * One "tick" = one execution of the body of
 *     ZMPVelocityReferencedQP::OnLine            (ZMPVelocityReferencedQP.cpp:346-452)
 * for one gait: support-state preview, orientation preview, QP assembly,
 * QL solve, LIPM interpolation + state step, trunk and feet interpolation.
 * The reference keeps this state in C++ objects and four deques; here it is a
 * flat POD per gait so that B gaits sit in one HBM array.
 * ---------------------------------------------------------------------- */

/* FootAbsolutePosition without time/stepType (include/jrl/walkgen/pgtypes.hh:137-154);
 * x,y,z in metres, angles in DEGREES like the reference. */
typedef struct wg_foot_sample {
  double x, y, z, theta, omega, omega2;
  double dx, dy, dz, dtheta, domega, domega2;
  double ddx, ddy, ddz, ddtheta, ddomega, ddomega2;
} wg_foot_sample_t;

enum { WG_LEFT = 0, WG_RIGHT = 1 };   /* foot_type_e, privatepgtypes.hh:47-50 */
enum { WG_SS = 0, WG_DS = 1 };        /* PhaseType,   privatepgtypes.hh:65-68 */

/* Robot / algorithm constants (what the path reads from CjrlHumanoidDynamicRobot
 * plus the constants hard-coded in the reference's constructors). */
typedef struct wg_model {
  int N;                    /* QP_N_ = 16            ZMPVelocityReferencedQP.cpp:64  */
  int flags;                /* WG_FLAG_* below */
  double T;                 /* QP_T_ = 0.1                                     :63  */
  double Tctrl;             /* m_SamplingPeriod = 0.005                        :65  */
  double com_height_qp;     /* 0.814                                      :103,114  */
  double alpha, beta, gamma;/* 1.0, 1e-5, 1e-6 (INSTANT_VELOCITY, JERK_MIN, COP_CENTERING) :116-118 */
  double sole_w, sole_h;    /* CjrlFoot::getSoleSize outputs, relative-feet-inequalities.cpp:158-172 */
  double margin_x, margin_y;/* 0.04, 0.04                                      :48-49 */
  double ds_feet_distance;  /* 0.2                                             :47   */
  double hip_l_lo, hip_l_hi, hip_r_lo, hip_r_hi; /* hip-yaw limits, OrientationsPreview.cpp:42-66 */
  double hip_vmax;          /* |upperVelocityBound|                            :68   */
  double hip_amax;          /* 0.1                                             :71   */
  double feet_cross_max;    /* 5 deg in rad                                    :73   */
  double step_period;       /* 0.8   SupportFSM, ZMPVelocityReferencedQP.cpp:76     */
  double ds_period;         /* 1e9                                             :77   */
  double dsss_period;       /* 0.8                                             :78   */
  double t_single;          /* 0.7   rigid-body-system.cpp:60                        */
  double t_double;          /* = T   rigid-body-system.cpp:61                        */
  double step_height;       /* 0.05  rigid-body-system.cpp:66                        */
  double feet_distance;     /* 0.2   rigid-body-system.cpp:65                        */
} wg_model_t;

/* wg_model_t.flags */
#define WG_FLAG_NO_STOP_CENTERING 1  /* skip the "go to the feet centre when NbStepsLeft == 0" branch
                                        (ZMPVelocityReferencedQP.cpp:410-421).  That branch was added
                                        after the reference's golden .datref was recorded (ChangeLog
                                        [3.1.8] "Put the CoM at the center of the feet when stopping");
                                        the flag exists so the golden file can be replayed. */
/* Where wg_mpc_configure takes the invariant Hessian block Q_b = beta I + alpha Uv'Uv + gamma Uz'Uz from
 * (GeneratorVelRef::build_invariant_part, generator-vel-ref.cpp:587-614).  Default (neither flag): the host loop in the
 * reference's summation order -- the tick is then bit-exact.  With a flag: the matrix cores (wg_gramian_batch), in fp64
 * (entries equal to rounding, 1e-14) or in fp32 (operands rounded to float: BASELINE's "fp32, MFMA Gramian condensation
 * path"; entries to 8e-8 of the largest one, which moves the jerk solution by ~1e-3 relative at N = 32 because Q_b's
 * smallest eigenvalue is beta = 1e-5 -- a tolerance mode, not a parity mode). */
#define WG_FLAG_GRAMIAN_MFMA_F64 2
#define WG_FLAG_GRAMIAN_MFMA_F32 4

/* Per-gait state carried from tick to tick. */
typedef struct wg_gait_state {
  /* scheduling: ZMPVelocityReferencedQP members + the caller's 5 ms clock */
  double clock;             /* m_InternalClock at which the NEXT tick fires */
  double upper_time_limit;  /* UpperTimeLimitToUpdate_ */
  double time_to_stop;      /* TimeToStopOnLineMode_ */
  int tick_count;
  int running;              /* Running_ */
  int ending_phase;         /* EndingPhase_ (":stoppg") */
  int online;               /* m_OnLineMode */
  /* NewVelRef_.Local */
  double vref[3];
  /* LinearizedInvertedPendulum2D state; com_z = starting CoM height (:303) */
  double com_x[3], com_y[3], com_z;
  /* IntermedData_->SupportState()  (support_state_t, privatepgtypes.hh:291-320) */
  int phase, foot, nb_steps_left, step_number, state_changed, pad0_;
  double time_limit, start_time, sup_x, sup_y, sup_yaw;
  /* SupportFSM rotation bookkeeping (SupportFSM.hh:117-134) */
  int in_translation, in_rotation, nb_steps_after_rotation, rot_support_foot,
      post_rotation_phase, nb_steps_ssds;
  /* OrientationsPreview::TrunkState_ / TrunkStateT_ (yaw only) */
  double trunk_yaw[3], trunkT_yaw[3];
  /* what the tick reads from the four deques: with the reference's cadence
   * (20 samples pushed per tick, one popped per 5 ms call) deque.front() is the
   * previous tick's sample #11, deque[size-2] its #18 and deque.back() its #19 */
  wg_foot_sample_t lf[3], rf[3];          /* [0] front, [1] back-1, [2] back */
  double front_com_x[3], front_com_y[3];  /* FinalCOMTraj_deq[0] */
  /* swing-foot height polynomial, set only when the support state changes
   * (OnLineFootTrajectoryGeneration.cpp:285-286) */
  double poly_z[5];
} wg_gait_state_t;

#define WG_SAMPLES_PER_TICK 20   /* QP_T_ / m_SamplingPeriod */

/* What one tick appends to the four deques + solver diagnostics.  Arrays of it should start on a 128-byte boundary
 * (hipMalloc / wg_host_alloc do). */
typedef struct wg_tick_out {
  double jerk_x, jerk_y;                       /* control applied to the LIPM */
  int ifail, n_iter, nact, n, m, nb_prw_steps; /* QP dimensions: n = 2N+2s, m = 1+4N+5s */
  double com_x[WG_SAMPLES_PER_TICK][3], com_y[WG_SAMPLES_PER_TICK][3];
  double com_yaw[WG_SAMPLES_PER_TICK][2];
  double zmp_x[WG_SAMPLES_PER_TICK], zmp_y[WG_SAMPLES_PER_TICK];
  wg_foot_sample_t lf[WG_SAMPLES_PER_TICK], rf[WG_SAMPLES_PER_TICK];
  /* the newest sample ALREADY in the feet queues before this tick: the double-support branch of
   * interpolate_feet_positions rewrites it (OnLineFootTrajectoryGeneration.cpp:333-336, k = 0) */
  wg_foot_sample_t lf_back, rf_back;
  /* sizeof == 7808 = 61 cache lines of 128 B (ABI 5; it was 7688, 8 B past 60 lines): in an array of these no line is shared
   * by two gaits' structs.  The tick writes zeros here, so the struct's bytes do not depend on what the buffer held.  (The
   * write amplification of an outs-on launch -- 11.3 KB leave the L2 per stored struct -- is the strided 8-byte stores of this
   * array-of-structs layout, not shared lines: DESIGN 2.) */
  double pad_[15];
} wg_tick_out_t;

/* Defaults of every constant the reference hard-codes; robot part = jrl-dynamics' sample robot. */
void wg_model_defaults(wg_model_t *model);

/* State after ZMPVelocityReferencedQP::InitOnLine (:212-319): standing in double
 * support on the left foot, queues holding the 8 start samples. */
void wg_gait_init(const wg_model_t *model, wg_gait_state_t *state,
                  const double com0[3] /* x y z */, const double left_xyt[3],
                  const double right_xyt[3] /* x y theta[deg] */);


/* Upload the model and build, on the device, what the reference builds once per gait in the
 * ZMPVelocityReferencedQP constructor and InitOnLine: the condensed cart-table maps S/U
 * (RigidBodySystem::compute_dyn_cjerk, rigid-body-system.cpp:377-452) and the invariant Hessian block
 * beta*I + alpha*Uv'Uv + gamma*Uz'Uz (GeneratorVelRef::build_invariant_part, generator-vel-ref.cpp:587-614).
 * These tables are constants of the model (a few KiB, 3 x 2N^3 flop once per process); they are formed on
 * the host in the reference's exact summation order -- so that the per-tick QPs are bit-identical to the
 * reference's -- and uploaded once. */
int wg_mpc_configure(const wg_model_t *model);
/* Optional: allocate the tick / run kernels' per-block solver slots and work queue for fleets of up to max_gaits now.  Without
 * it every launch sizes them for its own grid (a one-robot caller pays for one slot, not for a fleet's; growing never frees
 * what a launch in flight may be using). */
int wg_mpc_reserve(int max_gaits);

/* One MPC tick for each of B gaits (replaces B calls of ZMPVelocityReferencedQP::OnLine with
 * time + 1e-5 > UpperTimeLimitToUpdate_, ZMPVelocityReferencedQP.cpp:346-452).
 *   states  B structs, read and written; state->clock must hold the control-loop time of the tick
 *           (`advance_calls` > 0 adds that many control periods to it first, the way
 *           RunOneStepOfTheControlLoop does, PatternGeneratorInterfacePrivate.cpp:1256)
 *   outs    B structs or NULL (NULL: only the state is advanced)
 *   diag    B x 6 ints {ifail, n_iter, nact, n, m, nb_prw_steps} or NULL
 *   hist    B x hist_cap active-set add(+)/drop(-) log or NULL, hist_len B or NULL
 * The kernels keep a few solver arrays per block in a buffer of the context (what does not fit the CU's LDS at the
 * residency they run at), so launches of the tick / run entry points ON ONE CONTEXT do not overlap on the device.  The library
 * sees to that itself: a launch that arrives on a different stream while the context's previous tick / run launch has not
 * completed is enqueued BEHIND it (hipStreamWaitEvent on an event the previous launch left) -- a pipeline that orders its
 * streams with events of its own is accepted as it is, an unordered one is serialised instead of corrupted.  Streams that
 * are meant to overlap take one context each (wg_ctx_create).  With WG_OVERLAP_STRICT=1 in the environment when the context is
 * created (or after wg_set_overlap_strict(1)) such a launch is refused instead (WG_ERR_BUSY, nothing launched): a way to find
 * serialisation that was not intended; wg_overlap_serialised() counts the launches that were ordered.  The ordering test, the
 * launch and the event record are one critical section per context (host threads may share a context).
 * With more gaits than the device keeps resident, a call that follows another one on the same `states` array starts the gaits
 * in the order of decreasing QL iteration count of that previous tick (scheduling only: no result depends on it). */
int wg_mpc_tick_batch(int B, wg_gait_state_t *states, wg_tick_out_t *outs, int *diag, int advance_calls,
                      int *hist, int hist_cap, int *hist_len);
int wg_mpc_tick_batch_dev(int B, wg_gait_state_t *states, wg_tick_out_t *outs, int *diag, int advance_calls,
                          int *hist, int hist_cap, int *hist_len, void *hip_stream);
/* n_ticks consecutive ticks of every gait in ONE launch (fleet simulation, Monte-Carlo gaits).  A gait's tick t+1
 * depends only on its own tick t, so the batch does not synchronise between ticks: resident waves pull (gait, next
 * tick) items from a device-side queue, and a gait is offered again as soon as its tick is done.  Same results as
 * n_ticks calls of wg_mpc_tick_batch_dev with the same advance_calls, bit for bit; the velocity references stay what they
 * are for the whole launch (change them between launches with wg_mpc_set_velref_dev).
 *   outs  n_ticks x B (tick-major) or NULL;  diag  n_ticks x B x 6 (tick-major) or NULL.
 * No add/drop history in this mode.  The queue lives in a buffer of the context: one such launch in flight per context at
 * a time (launches on one stream are fine -- they run one after the other). */
int wg_mpc_run_batch_dev(int B, wg_gait_state_t *states, int n_ticks, int advance_calls, wg_tick_out_t *outs, int *diag,
                         void *hip_stream);
/* The same with the velocity references staged ahead: before tick t = 0, period, 2 period, ... of the launch, gait g's
 * reference becomes vref_sched[(t / period) * 3 B + 3 g + {0, 1, 2}] (DEVICE pointer, ceil(n_ticks / period) x B x 3
 * doubles) -- exactly what a loop of { wg_mpc_set_velref_dev; wg_mpc_run_batch_dev(period ticks) } does, bit for bit, in one
 * launch: the batch does not drain at every change of the references.  The batched form of a caller that issues
 * ":setVelReference" / setVelocityReference (patterngeneratorinterface.hh:291-293, ZMPVelocityReferencedQP.hh:103-114)
 * every `period` ticks of the control loop.  period >= 1; vref_sched == NULL: wg_mpc_run_batch_dev. */
int wg_mpc_run_sched_dev(int B, wg_gait_state_t *states, int n_ticks, int advance_calls, const double *vref_sched,
                         int period, wg_tick_out_t *outs, int *diag, void *hip_stream);
/* NewVelRef_ <- (vx, vy, vyaw) for every gait (":setVelReference", ZMPVelocityReferencedQP.hh:103-114);
 * vref = B x 3 doubles, DEVICE pointers. */
int wg_mpc_set_velref_dev(int B, wg_gait_state_t *states, const double *vref, void *hip_stream);
/* One robot (BASELINE configs[1]: batch = 1 -- what a drop-in caller of ZMPVelocityReferencedQP::OnLine has, one tick per
 * 0.1 s of walking): the same tick with the caller's state, outputs and diagnostics in HOST-MAPPED memory from
 * wg_host_alloc.  No staging copies and no device synchronisation: the kernel reads the state over the bus, works on a device
 * copy, writes state / out / diag back and releases a completion counter the call spins on.  Same bytes as
 * wg_mpc_tick_batch(1, ...); out and diag may be NULL.  Runs on a stream of the context, ordered behind the context's other
 * tick / run launches like any launch on another stream (above). */
int wg_host_alloc(void **out, size_t bytes);
void wg_host_free(void *p);
int wg_mpc_tick_pinned(wg_gait_state_t *state, wg_tick_out_t *out, int *diag, int advance_calls);
/* The QP of every gait's NEXT tick exactly as QPProblem::solve would hand it to ql0001_ (qp-problem.cpp:245-279), without
 * advancing anything: Q (C), D (d), DU (A, row 0 the dummy row, mmax >= m + 1 rows), DS (b), XL, XU -- the arrays
 * QPProblem::dump_problem prints (qp-problem.cpp:639-653), which the reference writes to /tmp/Problem_<time>.dat when a solve
 * fails (ZMPVelocityReferencedQP.cpp:399-402).  Column-major with the caller's leading dimensions (nmax, mmax), zero-padded;
 * n[g], m[g] receive the sizes (m counts the dummy row, as m_ does).  `advance_calls` as in wg_mpc_tick_batch.  The states are
 * not modified.  Feeding the result to wg_qp_solve_batch gives the x the fused tick computes, bit for bit. */
int wg_mpc_assemble_batch(int B, const wg_gait_state_t *states, int advance_calls, int nmax, int mmax, double *C, double *d,
                          double *A, double *b, double *xl, double *xu, int *n, int *m);
int wg_mpc_assemble_batch_dev(int B, const wg_gait_state_t *states, int advance_calls, int nmax, int mmax, double *C, double *d,
                              double *A, double *b, double *xl, double *xu, int *n, int *m, void *hip_stream);
/* LDS bytes one gait occupies in the tick kernel for the configured model. */
size_t wg_mpc_tick_lds_bytes(void);
/* the same figure for any model, without configuring it (host arithmetic: lets callers and tests reason about
 * residency -- a gfx950 CU has 160 KiB of LDS, handed out in 1280-byte granules); 0 if N is out of range */
size_t wg_mpc_tick_lds_bytes_for(const wg_model_t *model);

/* Kajita preview-control gains (host, no device work) -----------------------
 *
 * wg_riccati_solve replaces
 *     OptimalControllerSolver(A, b, c, Q, R, Nl) + ComputeWeights(mode) + GetK/GetF
 *     src/PreviewControl/OptimalControllerSolver.hh:143-144, OptimalControllerSolver.cpp:97-126, 200-352
 * for a single-input single-output system x+ = A x + b u, p = c x of order n <= 8 (A row-major n x n):
 * P = stabilising solution of the discrete Riccati equation with weights Q (output) and R (input),
 *     K[n]  = (R + b'Pb)^-1 b'PA,
 *     F[Nl] = (R + b'Pb)^-1 b' ((A - bK)')^k  (c'Q            mode WG_RICCATI_WITH_INITIALPOS
 *                                              P c'Q          mode WG_RICCATI_WITHOUT_INITIALPOS).
 * wg_riccati_gains replaces PreviewControl::ComputeOptimalWeights (src/PreviewControl/PreviewControl.cpp:198-322):
 * it builds the cart-table system for sampling period T and CoM height zc (g = 9.81) -- 3 states for
 * WITH_INITIALPOS, the 4-state error system for WITHOUT_INITIALPOS -- and returns
 *     WITHOUT_INITIALPOS: K = {Ks, Kx[0], Kx[1], Kx[2]};  WITH_INITIALPOS: K = {Kx[0] (= Ks), Kx[1], Kx[2], 0}.
 * The reference passes Q = 1, R = 1e-6 (WITHOUT) or 1e-5 (WITH), Nl = (int)(preview time / T). */
#define WG_RICCATI_WITH_INITIALPOS 0
#define WG_RICCATI_WITHOUT_INITIALPOS 1
int wg_riccati_solve(int n, const double *A, const double *b, const double *c, double Q, double R, int Nl, int mode,
                     double *K, double *F);
int wg_riccati_gains(double T, double zc, double Q, double R, int Nl, int mode, double *K, double *F);

/* PLDP: primal active-set solver in LQ-preconditioned coordinates (Dimitrov 2008) with OptCholesky updates ----------
 *
 * wg_pldp_configure replaces the constructor
 *     PLDPSolver(CardU, iPu, Px, Pu, iLQ)            src/Mathematics/PLDPSolver.cpp:40-96 (+ PrecomputeiPuPx :208-285)
 * and wg_pldp_solve_batch replaces, batched over B independent problems, each with its own hot-start state,
 *     PLDPSolver::SolveProblem(CstPartOfTheCostFunction, NbOfConstraints, LinearPartOfConstraints,
 *                              CstPartOfConstraints, ZMPRef, XkYk, X, SimilarConstraint,
 *                              NumberOfRemovedConstraints, StartingSequence)      PLDPSolver.cpp:654-1007
 * including the row-append Cholesky of E E' it drives,
 *     OptCholesky::AddActiveConstraint / UpdateCholeskyMatrixFortran              OptCholesky.cpp:92-104, 171-223
 * called from ZMPConstrainedQPFastFormulation::BuildZMPTrajectoryFromFootTrajectory
 *     src/ZMPRefTrajectoryGeneration/ZMPConstrainedQPFastFormulation.cpp:1318-1327.
 *
 *     minimise 1/2 |v|^2 + D'v   s.t.  A v + b >= 0,      v in R^(2N)  (x block, then y block)
 *
 *   N        preview length (CardU), 1 <= N <= WG_PLDP_N;  iPu, Pu: N x N row-major, Px: N x 3 row-major
 *   mcap     slot size of the per-problem arrays, m[b] <= mcap <= WG_PLDP_MMAX
 *   m        B          number of constraint rows of problem b
 *   D        B x 2N     linear part of the cost
 *   A        B slots of (mcap+1)*2N doubles; problem b is COLUMN-major with LEADING DIMENSION m[b]+1, exactly as
 *                       BuildConstraintMatrices writes DPu (ZMPConstrainedQPFastFormulation.cpp:773-775, 893-905)
 *   b        B x mcap   constant part (DPx)
 *   zmpref   B x 2N,  xkyk B x 6 (x, dx, ddx, y, dy, ddy)
 *   similar  B x mcap   SimilarConstraint: 0, or the (negative) offset to an earlier row with A_i = -A_j
 *                       (FootConstraintsAsLinearSystem.cpp:55-93); positive offsets are rejected
 *   n_removed, starting   B each: NumberOfRemovedConstraints, StartingSequence
 *   max_iter  the reference stops on a 1.3 ms wall-clock budget (PLDPSolver.cpp:51-52, 889-900), which cannot be
 *             reproduced; max_iter <= 0 runs to completion (the reference when the budget is not hit), otherwise the
 *             loop ends after max_iter iterations, like a budget that expires during that iteration
 *   states   B          hot-start state, read and updated (m_PreviouslyActivatedConstraints, m_PreviousZMPSolution)
 *   X        B x 2N     solution;   ret B: 0, WG_PLDP_NAN (reference returns -1), WG_PLDP_NEG_ALPHA (reference calls
 *                       exit(0)), WG_PLDP_CAPACITY (more than WG_PLDP_ACTIVE_CAP active rows; E E' is singular long
 *                       before that), WG_PLDP_BAD_INPUT (a positive or out-of-range SimilarConstraint offset)
 *   n_iter   B          m_ItNb;  active B x mcap / n_active B: m_ActivatedConstraints in activation order (or NULL) */
#define WG_PLDP_N 16
#define WG_PLDP_MMAX (8 * WG_PLDP_N)
/* rows the packed Cholesky factor L of E E' holds.  With 2N = 32 unknowns E E' is singular beyond 32 active rows (at most two
 * faces of a ZMP polygon meet per instant), so 40 is margin, not a limit of the method -- and L is what decides how many problems
 * a CU holds (64 rows: 21 KB of LDS per problem, 7 per CU; 40: 10.9 KB, 14 per CU: +36 % solves/s).  The Dimitrov tick kernel
 * uses the same cap. */
#define WG_PLDP_ACTIVE_CAP 40
#define WG_PLDP_NAN (-1)
#define WG_PLDP_NEG_ALPHA (-2)
#define WG_PLDP_CAPACITY (-3)
#define WG_PLDP_BAD_INPUT (-4)
typedef struct wg_pldp_state {
  int n_prev;
  int prev_active[WG_PLDP_MMAX];
  int pad_;
  double prev_zmp[2 * WG_PLDP_N];
  double internal_time;
} wg_pldp_state_t;
int wg_pldp_configure(int N, const double *iPu, const double *Px, const double *Pu);
int wg_pldp_solve_batch(int B, int mcap, const int *m, const double *D, const double *A, const double *b,
                        const double *zmpref, const double *xkyk, const int *similar, const int *n_removed,
                        const int *starting, int max_iter, wg_pldp_state_t *states, double *X, int *ret, int *n_iter,
                        int *active, int *n_active);
/* same, DEVICE pointers, asynchronous on hip_stream */
int wg_pldp_solve_batch_dev(int B, int mcap, const int *m, const double *D, const double *A, const double *b,
                            const double *zmpref, const double *xkyk, const int *similar, const int *n_removed,
                            const int *starting, int max_iter, wg_pldp_state_t *states, double *X, int *ret,
                            int *n_iter, int *active, int *n_active, void *hip_stream);
/* LDS bytes one problem occupies in the PLDP kernel. */
size_t wg_pldp_lds_bytes(void);

/* Dimitrov-2008 receding-horizon tick around PLDP, batched over independent gaits ------------------------------------
 *
 * wg_dimitrov_configure replaces ZMPConstrainedQPFastFormulation::InitConstants
 *     src/ZMPRefTrajectoryGeneration/ZMPConstrainedQPFastFormulation.cpp:692-703  (= InitializeMatrixPbConstants :158-246,
 *     BuildingConstantPartOfTheObjectiveFunction[QLDANDLQ] :384-560, BuildingConstantPartOfConstraintMatrices :597-690)
 * in mode PLDP (:55) and hands the resulting iPu, Px, Pu to the PLDP back-end (the PLDPSolver constructor, :104-109).
 * wg_dimitrov_tick_batch replaces one pass of the loop body of BuildZMPTrajectoryFromFootTrajectory (:1180-1400):
 *     BuildConstraintMatrices (:759-1022)  from the ZMP polytopes of the N previewed instants
 *     D = OptB xk - OptC ZMPRef (:1254-1262), PLDPSolver::SolveProblem (:1322-1339), X <- iLQ' X (:1355-1381),
 *     LinearizedInvertedPendulum2D::Interpolation + OneIteration (:1388-1392).
 * The polytopes are what FootConstraintsAsLinearSystem::BuildLinearConstraintInequalities produces
 * (LinearConstraintInequality_t, pgtypes.hh:158-165): rows A_j (2 coefficients), B_j, the centre (the ZMP reference)
 * and SimilarConstraints; constraint sense A_j . zmp + B_j >= 0.  The caller supplies, per gait, the polytope of each
 * previewed instant (N entries) -- 2 KB per gait-tick instead of the 33 KB dense DPu. */
#define WG_POLY_MAX_ROWS 8            /* "8 constraints per support foot", :771-772 */
/* the QP back-end of the tick, m_FastFormulationMode (ZMPConstrainedQPFastFormulation.hh:263-265; the constructor takes PLDP, :55) */
#define WG_DIMITROV_PLDP 0            /* PLDPSolver on the LQ-preconditioned problem (v = LQ u, Hessian = I), X <- iLQ' X            */
#define WG_DIMITROV_QLD 1             /* "QLD" (:1297-1320): ql0001_ on the problem as it stands -- Q = OptA (:382-389), constraint  */
                                      /* matrix from the triangular Pu' (:891-904), cost vector from OptB / OptC as built (:545-556), */
                                      /* iwar[0] = 1, eps = 1e-8, bounds +-1e8 (:1280-1292); X is the jerk itself (:1383).  With the   */
                                      /* reference's own OptA (alpha VPu' instead of alpha VPu'VPu, :524-527: not symmetric, its upper  */
                                      /* triangle not positive definite) ql0001_ answers ifail = 2 on the first tick -- here as there. */
#define WG_DIMITROV_QLDANDLQ 2        /* "QLDANDLQ": ql0001_ on the SAME preconditioned problem PLDP gets -- Q = I handed over as its   */
                                      /* own Cholesky factor (iwar[0] = 0, :1288-1289), DPu from iLQ Pu' (:905-918), X <- iLQ' X        */
typedef struct wg_dimitrov_model {
  int N, solver;                      /* m_QP_N = 16 (:83); WG_DIMITROV_PLDP (0, the reference's default), _QLD or _QLDANDLQ */
  double T;                           /* m_QP_T = 0.1                                 :82  */
  double Tctrl;                       /* m_SamplingPeriod = 0.005 */
  double com_height;                  /* m_ComHeight = 0.80                           :87  */
  double alpha, beta;                 /* 200, 1000                                    :95-96 */
} wg_dimitrov_model_t;
typedef struct wg_zmp_polytope {
  int nrows, pad_;
  int similar[WG_POLY_MAX_ROWS];
  double A[WG_POLY_MAX_ROWS][2];
  double B[WG_POLY_MAX_ROWS];
  double centre[2];
} wg_zmp_polytope_t;
typedef struct wg_dimitrov_state {
  double xk[6];                       /* LIPM state x, dx, ddx, y, dy, ddy  (m_2DLIPM->GetState) */
  wg_pldp_state_t pldp;               /* the solver's hot-start members */
  int n_removed;                      /* NumberOfRemovedConstraints for the next solve (:1340) */
  int starting;                       /* StartingSequence: 1 before the first tick (:1167, :1339) */
} wg_dimitrov_state_t;
typedef struct wg_dimitrov_out {
  double jerk_x, jerk_y;              /* ptX[0], ptX[N] */
  int ret, n_iter, n_active, m;       /* PLDP return code (see wg_pldp_solve_batch), m_ItNb, active rows, rows;
                                       * WG_DIMITROV_QLD: ql0001_'s ifail (0 = solved), its iteration count, active constraints, rows */
  /* what Interpolation writes for lk = 0..interval (21 samples; the last one is overwritten by the next tick) */
  double com_x[WG_SAMPLES_PER_TICK + 1][3], com_y[WG_SAMPLES_PER_TICK + 1][3];
  double zmp_x[WG_SAMPLES_PER_TICK + 1], zmp_y[WG_SAMPLES_PER_TICK + 1];
  double X[2 * WG_PLDP_N];            /* ptX: the jerks over the horizon (x block, then y block), un-preconditioned (:1355-1383) */
} wg_dimitrov_out_t;
void wg_dimitrov_defaults(wg_dimitrov_model_t *model);
int wg_dimitrov_configure(const wg_dimitrov_model_t *model);
/* the constants InitConstants leaves behind, for inspection: iLQ, OptC 2N x 2N, OptB 2N x 6, Pu (= iLQ Pu'), iPu N x N,
 * Px N x 3, all row-major; any pointer may be NULL */
int wg_dimitrov_get_constants(double *iLQ, double *OptB, double *OptC, double *Pu, double *iPu, double *Px);
/* what mode QLD works with: Q = m_Q (2N x 2N, ql0001_'s column-major layout, :382-389), OptB (2N x 6) and OptC (2N x 2N) as
 * BuildingConstantPartOfTheObjectiveFunction leaves them before the LQ step (:545-556), PuT = Pu' (N x N, [k N + i], :629-637);
 * row-major unless said otherwise; any pointer may be NULL */
int wg_dimitrov_get_qld_constants(double *Q, double *OptB, double *OptC, double *PuT);
/* polys: B x N polytopes (instant-major per gait); outs may be NULL.  A gait whose solve returns ret != 0 keeps its
 * state (xk not advanced), like the reference which stops there (:1343-1347).  With model.solver == WG_DIMITROV_QLD / _QLDANDLQ the
 * solve is the in-wave ql0002 of wg_qp_solve_batch -- the back-ends of this tick whose CPU counterpart is the reference's own
 * compiled qld.cpp; the PLDP hot-start members of the state are carried along untouched, max_iter is ignored.  With more gaits
 * than resident waves they start longest-solve-first by the previous tick on the same state array (scheduling only). */
int wg_dimitrov_tick_batch(int B, const wg_zmp_polytope_t *polys, wg_dimitrov_state_t *states, wg_dimitrov_out_t *outs,
                           int max_iter);
int wg_dimitrov_tick_batch_dev(int B, const wg_zmp_polytope_t *polys, wg_dimitrov_state_t *states,
                               wg_dimitrov_out_t *outs, int max_iter, void *hip_stream);

/* Kajita stage-1 preview control, batched over independent gaits ------------------------------------------------------
 *
 * wg_preview_configure replaces the members PreviewControl holds once ComputeOptimalWeights
 * (src/PreviewControl/PreviewControl.cpp:198-322) or ReadPrecomputedFile (:134-194) has run: sampling period T and CoM
 * height zc (=> m_A, m_B, m_C :203-214), m_Kx, m_Ks and the window gains m_F[nl], nl = (int)(preview time / T).
 * (wg_riccati_gains computes them; mode WITHOUT_INITIALPOS returns K = {Ks, Kx[0..2]}.)
 * wg_preview_run_batch replaces, for each of B independent gaits, L consecutive calls
 *     PreviewControl::OneIterationOfPreview(x, y, sxzmp, syzmp, ZMPPositions, lindex = l, zmpx2, zmpy2, Simulation)
 *     src/PreviewControl/PreviewControl.cpp:324-374        (OneIterationOfPreview1D :376-420 is one axis of it)
 * for l = 0..L-1 on a ZMP reference queue of L + nl - 1 samples (the reference throws when fewer than nl samples are
 * left, :341-344; here the caller sizes the queue).  Same operation order as the reference, bit for bit.
 *   zmp_x, zmp_y   B x (L + nl - 1)   ZMPPositions[.].px / .py per gait
 *   state          B x 8              x(0..2,0), y(0..2,0), sxzmp, syzmp -- read, advanced L steps, written back
 *   com            B x L x 6          x and y after each step (what callers push on the CoM buffer), may be NULL
 *   zmp2           B x L x 2          zmpx2, zmpy2 of each step, may be NULL
 *   simulation     the reference's `Simulation` flag: accumulate the ZMP tracking error into sxzmp / syzmp
 * The _dev variant takes TIME-MAJOR device arrays so that a wave's window reads are contiguous:
 *   zmp_*_tm [(L + nl - 1)][B],  com_tm [L][6][B],  zmp2_tm [L][2][B];  state stays [B][8]. */
#define WG_PREVIEW_NL_MAX 4096
typedef struct wg_preview_gains {
  double T, zc;                       /* m_SamplingPeriod, m_Zc */
  double Ks, Kx[3];                   /* m_Ks, m_Kx */
  int nl, pad_;                       /* m_SizeOfPreviewWindow */
} wg_preview_gains_t;
int wg_preview_configure(const wg_preview_gains_t *gains, const double *F);
int wg_preview_window(void);          /* nl of the configured gains, 0 before wg_preview_configure */
int wg_preview_run_batch(int B, int L, const double *zmp_x, const double *zmp_y, double *state, double *com, double *zmp2,
                         int simulation);
int wg_preview_run_batch_dev(int B, int L, const double *zmp_x_tm, const double *zmp_y_tm, double *state, double *com_tm,
                             double *zmp2_tm, int simulation, void *hip_stream);

/* Kajita stage-1 inputs: ZMP reference queue and feet trajectories of a step sequence, batched over gaits ------------
 *
 * wg_zmpdisc_batch replaces, for each of B independent gaits,
 *     ZMPDiscretization::GetZMPDiscretization                  src/ZMPRefTrajectoryGeneration/ZMPDiscretization.cpp:143-173
 * = InitOnLine (:319-513: rest phase of 2 preview windows, then OnLineAddFoot (:573-1020) for every step after the
 * first) + EndPhaseOfTheWalking (:1129-1300), i.e. what PatternGeneratorInterfacePrivate::CreateZMPReferences
 * (PatternGeneratorInterfacePrivate.cpp:1870-1882) obtains for ":stepseq" in Kajita mode -- with
 *     FilterOutValues (:1045-1109) on the window of InitializeFilter (:240-262),
 *     UpdateCurrentSupportFootPosition (:515-558),
 *     FootTrajectoryGenerationStandard::UpdateFootPosition / SetParameters
 *         src/FootTrajectoryGeneration/FootTrajectoryGenerationStandard.cpp:411-566, 150-186,
 *     Polynome::Compute, Polynome3/4/5::SetParameters   src/Mathematics/Polynome.cpp:44-53, PolynomeFoot.cpp:41-57, 100-120, 174-195.
 * The steps are RelativeFootPosition records (pgtypes.hh:100-108) as StepStackHandler::ReadStepSequenceAccordingToWalkMode
 * (StepStackHandler.cpp:128-175) leaves them: theta in DEGREES, SStime / DStime = the configured support times,
 * stepType 1.  Output per gait: L = wg_zmpdisc_length(model, steps, n_steps) samples at period T of
 *     the filtered ZMP reference (px, py) -- the queue PreviewControl::OneIterationOfPreview reads,
 *     its heading theta and stepType, and both feet (x, y, z, theta, omega, omega2, stepType).
 *   steps      B x smax  (gait b uses the first n_steps[b]; 2 <= n_steps[b] <= smax <= WG_ZMPDISC_MAX_STEPS)
 *   init_feet  B x 6     left x, y, theta, right x, y, theta at the start (EvaluateStartingState's outputs)
 *   lcap       row length of the per-gait output arrays (>= every gait's L; samples past L are left untouched)
 *   zmp        B x lcap x 2  (px, py);  zmp_theta B x lcap;  zmp_type B x lcap   (may be NULL each)
 *   left, right  B x lcap x 6;  left_type, right_type  B x lcap                      (may be NULL each)
 *   length     B         L of each gait, or a negative code: WG_ZMPDISC_BAD_INPUT (n_steps out of range, a phase that
 *                        does not fit its own sample count -- the reference would write out of bounds), WG_ZMPDISC_CAPACITY
 * The _dev variant takes device pointers and writes the ZMP queue TIME-MAJOR, zmp_x_tm / zmp_y_tm [lcap][B], exactly
 * what wg_preview_run_batch_dev reads, so that step sequences go to CoM trajectories without leaving the device; samples
 * past a gait's L repeat its last value there (a gait at rest), so that one preview launch can cover a ragged batch. */
#define WG_ZMPDISC_MAX_STEPS 64
#define WG_ZMPDISC_BAD_INPUT (-1)
#define WG_ZMPDISC_CAPACITY (-2)
typedef struct wg_zmpdisc_model {
  double T;                           /* m_SamplingPeriod  0.005 */
  double preview_time;                /* m_PreviewControlTime  1.6 */
  double t_single, t_double;          /* m_Tsingle, m_Tdble  (":singlesupporttime", ":doublesupporttime") */
  double step_height;                 /* m_StepHeight  (":stepheight") */
  double omega;                       /* m_Omega, degrees  (":omega") */
  double modulation;                  /* m_ModulationSupportCoefficient = 0.9   ZMPDiscretization.cpp:99 */
  double zmp_neutral[2];              /* m_ZMPNeutralPosition = 0, 0            :107-108 */
  double zmp_shift[4];                /* m_ZMPShift (step types 3, 4, 5)        :103-105, 693-718 */
  double foot_b, foot_h, foot_f;      /* m_FootB, m_FootH, m_FootF  FootTrajectoryGenerationStandard.cpp:68-71 */
} wg_zmpdisc_model_t;
typedef struct wg_rel_step {
  double sx, sy, theta;               /* RelativeFootPosition, theta in degrees */
  double ss_time, ds_time;            /* SStime, DStime (ds_time == 0 => the model's times, :608-612) */
  int step_type, pad_;
} wg_rel_step_t;
void wg_zmpdisc_defaults(wg_zmpdisc_model_t *model);
/* number of samples GetZMPDiscretization produces for this sequence (host arithmetic), or a negative code */
int wg_zmpdisc_length(const wg_zmpdisc_model_t *model, const wg_rel_step_t *steps, int n_steps);
int wg_zmpdisc_batch(const wg_zmpdisc_model_t *model, int B, int smax, const wg_rel_step_t *steps, const int *n_steps,
                     const double *init_feet, int lcap, double *zmp, double *zmp_theta, int *zmp_type, double *left,
                     int *left_type, double *right, int *right_type, int *length);
int wg_zmpdisc_batch_dev(const wg_zmpdisc_model_t *model, int B, int smax, const wg_rel_step_t *steps,
                         const int *n_steps, const double *init_feet, int lcap, double *zmp_x_tm, double *zmp_y_tm,
                         int *length, void *hip_stream);
/* every output on the device, all time-major: zmp_theta_tm / types [lcap][B], left_tm / right_tm [lcap][6][B]
 * (x, y, z, theta, omega, omega2); any output pointer may be NULL */
int wg_zmpdisc_full_batch_dev(const wg_zmpdisc_model_t *model, int B, int smax, const wg_rel_step_t *steps,
                              const int *n_steps, const double *init_feet, int lcap, double *zmp_x_tm, double *zmp_y_tm,
                              double *zmp_theta_tm, int *zmp_type_tm, double *left_tm, int *left_type_tm, double *right_tm,
                              int *right_type_tm, int *length, void *hip_stream);

/* ZMP polytopes of a feet trajectory (host) ----------------------------------------------------------------------------
 *
 * wg_foot_constraints replaces FootConstraintsAsLinearSystem::BuildLinearConstraintInequalities
 *     src/Mathematics/FootConstraintsAsLinearSystem.cpp:258-539   (+ ComputeLinearSystem :97-256,
 *     FindSimilarConstraints :55-92, ComputeConvexHull::DoComputeConvexHull src/Mathematics/ConvexHull.cpp:88-203):
 * one polytope per support phase of a 5 ms feet trajectory, A_j . zmp + B_j >= 0, with its validity interval -- the
 * queue ZMPConstrainedQPFastFormulation::BuildConstraintMatrices (:759-1022) walks to fill the wg_zmp_polytope_t of each
 * previewed instant.  Host code: it runs once per step sequence, not per tick.
 *   n            samples;  time n;  left, right  n x 6 (x, y, z, theta, omega, omega2);  left_type n (stepType)
 *   sole_w, sole_h   getSoleSize outputs;  constraint_x, constraint_y  the security margins (ConstraintOnX / Y)
 *   polys, t_start, t_end   up to cap entries;  returns the number of polytopes (may exceed cap) or a negative code */
int wg_foot_constraints(int n, const double *time, const double *left, const int *left_type, const double *right,
                        double sole_w, double sole_h, double constraint_x, double constraint_y, int cap,
                        wg_zmp_polytope_t *polys, double *t_start, double *t_end);

/* Invariant Hessian block on the matrix cores, batched over models -----------------------------------------------------
 *
 * wg_gramian_batch computes, for B models that differ in QP sampling period T[b] and CoM height h[b],
 *     Q_b = beta I + alpha Uv'Uv + gamma Uz'Uz      (N x N, row-major, one block after the other)
 * i.e. GeneratorVelRef::build_invariant_part (src/ZMPRefTrajectoryGeneration/generator-vel-ref.cpp:587-614) on the maps
 * of RigidBodySystem::compute_dyn_cjerk (src/PreviewControl/rigid-body-system.cpp:377-452) -- the one GEMM-shaped step
 * of the path.  One wavefront per model; Uv and Uz are generated in registers from (T, h) and multiplied with
 * v_mfma_f64_16x16x4_f64 (WG_GRAMIAN_F64) or v_mfma_f32_16x16x4_f32 (WG_GRAMIAN_F32, operands rounded to float,
 * result widened).  The matrix cores fuse and reorder the sums: results match the host loop of wg_mpc_configure (which
 * keeps the reference's order, because the tick is bit-exact) to rounding only -- 1e-14 / 1e-6 relative. */
#define WG_GRAMIAN_F64 0
#define WG_GRAMIAN_F32 1
int wg_gramian_batch(int B, int N, const double *T, const double *h, double alpha, double beta, double gamma, int precision,
                     double *Qb);
int wg_gramian_batch_dev(int B, int N, const double *T, const double *h, double alpha, double beta, double gamma,
                         int precision, double *Qb, void *hip_stream);

/* The context forms of the entry points above (same arguments after the context, same semantics). ---------------------- */
int wg_qp_solve_batch_dev_ctx(wg_ctx_t *ctx, int B, int nmax, int mmax, const int *n, const int *m, const int *me,
                              const double *C, const double *d, const double *A, const double *b, const double *xl,
                              const double *xu, double eps, double *x, double *u, int *ifail, int *n_iter, int *iact,
                              int *nact, int *hist, int hist_cap, int *hist_len, void *hip_stream);
int wg_qp_solve_batch_ctx(wg_ctx_t *ctx, int B, int nmax, int mmax, const int *n, const int *m, const int *me,
                          const double *C, const double *d, const double *A, const double *b, const double *xl,
                          const double *xu, double eps, double *x, double *u, int *ifail, int *n_iter, int *iact,
                          int *nact, int *hist, int hist_cap, int *hist_len);
int wg_set_overlap_strict_ctx(wg_ctx_t *ctx, int on);
long long wg_overlap_serialised_ctx(wg_ctx_t *ctx);
int wg_mpc_configure_ctx(wg_ctx_t *ctx, const wg_model_t *model);
size_t wg_mpc_tick_lds_bytes_ctx(wg_ctx_t *ctx);
int wg_mpc_reserve_ctx(wg_ctx_t *ctx, int max_gaits);
int wg_mpc_tick_batch_dev_ctx(wg_ctx_t *ctx, int B, wg_gait_state_t *states, wg_tick_out_t *outs, int *diag,
                              int advance_calls, int *hist, int hist_cap, int *hist_len, void *hip_stream);
int wg_mpc_run_batch_dev_ctx(wg_ctx_t *ctx, int B, wg_gait_state_t *states, int n_ticks, int advance_calls,
                             wg_tick_out_t *outs, int *diag, void *hip_stream);
int wg_mpc_run_sched_dev_ctx(wg_ctx_t *ctx, int B, wg_gait_state_t *states, int n_ticks, int advance_calls,
                             const double *vref_sched, int period, wg_tick_out_t *outs, int *diag, void *hip_stream);
int wg_mpc_tick_batch_ctx(wg_ctx_t *ctx, int B, wg_gait_state_t *states, wg_tick_out_t *outs, int *diag,
                          int advance_calls, int *hist, int hist_cap, int *hist_len);
int wg_mpc_set_velref_dev_ctx(wg_ctx_t *ctx, int B, wg_gait_state_t *states, const double *vref, void *hip_stream);
int wg_mpc_tick_pinned_ctx(wg_ctx_t *ctx, wg_gait_state_t *state, wg_tick_out_t *out, int *diag, int advance_calls);
int wg_mpc_assemble_batch_ctx(wg_ctx_t *ctx, int B, const wg_gait_state_t *states, int advance_calls, int nmax, int mmax,
                              double *C, double *d, double *A, double *b, double *xl, double *xu, int *n, int *m);
int wg_mpc_assemble_batch_dev_ctx(wg_ctx_t *ctx, int B, const wg_gait_state_t *states, int advance_calls, int nmax, int mmax,
                                  double *C, double *d, double *A, double *b, double *xl, double *xu, int *n, int *m,
                                  void *hip_stream);
int wg_pldp_configure_ctx(wg_ctx_t *ctx, int N, const double *iPu, const double *Px, const double *Pu);
int wg_pldp_solve_batch_dev_ctx(wg_ctx_t *ctx, int B, int mcap, const int *m, const double *D, const double *A,
                                const double *b, const double *zmpref, const double *xkyk, const int *similar,
                                const int *n_removed, const int *starting, int max_iter, wg_pldp_state_t *states,
                                double *X, int *ret, int *n_iter, int *active, int *n_active, void *hip_stream);
int wg_pldp_solve_batch_ctx(wg_ctx_t *ctx, int B, int mcap, const int *m, const double *D, const double *A,
                            const double *b, const double *zmpref, const double *xkyk, const int *similar,
                            const int *n_removed, const int *starting, int max_iter, wg_pldp_state_t *states,
                            double *X, int *ret, int *n_iter, int *active, int *n_active);
int wg_dimitrov_configure_ctx(wg_ctx_t *ctx, const wg_dimitrov_model_t *model);
int wg_dimitrov_get_constants_ctx(wg_ctx_t *ctx, double *iLQ, double *OptB, double *OptC, double *Pu, double *iPu,
                                  double *Px);
int wg_dimitrov_get_qld_constants_ctx(wg_ctx_t *ctx, double *Q, double *OptB, double *OptC, double *PuT);
int wg_dimitrov_tick_batch_dev_ctx(wg_ctx_t *ctx, int B, const wg_zmp_polytope_t *polys, wg_dimitrov_state_t *states,
                                   wg_dimitrov_out_t *outs, int max_iter, void *hip_stream);
int wg_dimitrov_tick_batch_ctx(wg_ctx_t *ctx, int B, const wg_zmp_polytope_t *polys, wg_dimitrov_state_t *states,
                               wg_dimitrov_out_t *outs, int max_iter);
int wg_preview_configure_ctx(wg_ctx_t *ctx, const wg_preview_gains_t *gains, const double *F);
int wg_preview_window_ctx(wg_ctx_t *ctx);
int wg_preview_run_batch_dev_ctx(wg_ctx_t *ctx, int B, int L, const double *zmp_x_tm, const double *zmp_y_tm,
                                 double *state, double *com_tm, double *zmp2_tm, int simulation, void *hip_stream);
int wg_preview_run_batch_ctx(wg_ctx_t *ctx, int B, int L, const double *zmp_x, const double *zmp_y, double *state,
                             double *com, double *zmp2, int simulation);
int wg_gramian_batch_dev_ctx(wg_ctx_t *ctx, int B, int N, const double *T, const double *h, double alpha,
                             double beta, double gamma, int precision, double *Qb, void *hip_stream);
int wg_gramian_batch_ctx(wg_ctx_t *ctx, int B, int N, const double *T, const double *h, double alpha, double beta,
                         double gamma, int precision, double *Qb);
int wg_zmpdisc_batch_dev_ctx(wg_ctx_t *ctx, const wg_zmpdisc_model_t *model, int B, int smax,
                             const wg_rel_step_t *steps, const int *n_steps, const double *init_feet, int lcap,
                             double *zmp_x_tm, double *zmp_y_tm, int *length, void *hip_stream);
int wg_zmpdisc_full_batch_dev_ctx(wg_ctx_t *ctx, const wg_zmpdisc_model_t *model, int B, int smax,
                                  const wg_rel_step_t *steps, const int *n_steps, const double *init_feet, int lcap,
                                  double *zmp_x_tm, double *zmp_y_tm, double *zmp_theta_tm, int *zmp_type_tm,
                                  double *left_tm, int *left_type_tm, double *right_tm, int *right_type_tm,
                                  int *length, void *hip_stream);
int wg_zmpdisc_batch_ctx(wg_ctx_t *ctx, const wg_zmpdisc_model_t *model, int B, int smax, const wg_rel_step_t *steps,
                         const int *n_steps, const double *init_feet, int lcap, double *zmp, double *zmp_theta,
                         int *zmp_type, double *left, int *left_type, double *right, int *right_type, int *length);

#ifdef __cplusplus
}
#endif
#endif /* WG_MPC_H */
