/*
 * pyrovi.h -- C ABI of libpyrovi.so: MI355X (gfx950) grid value-iteration sweeps.
 *
 * The reference (SherbyRobotics/pyro) is pure Python and has no FFI boundary of its own; its
 * boundary is the Python class surface.  This header is the C boundary placed BEHIND that
 * surface: each entry point replaces one reference method (cited as file:line, relative to the
 * reference repo root).  The host-side mirror in pyro_amd/ binds these through ctypes
 * (pyro_amd/_native.py); INTEGRATION.md shows the equivalent stub a pyro maintainer would add.
 *
 * Conventions: plain pointers and sizes only; every function returns 0 on success or a negative
 * PVI_E* code (never throws across the ABI); pvi_last_error() returns a thread-local message.
 * Host buffers are borrowed for the duration of the call.  A handle owns its device memory and
 * one HIP stream and must be used from one host thread at a time.  Arrays follow the reference's
 * layout: node id / action id are C-order (last axis fastest) over x_dim / u_dim
 * (pyro/planning/discretizer.py:167-310); J is float64[nodes], pi is int64[nodes] on the host.
 */
#ifndef PYROVI_H
#define PYROVI_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define PVI_ABI_VERSION 3
#define PVI_MAX_N 4 /* state dimensions supported by the reference grid: 2, 3, 4 (discretizer.py:183-245) */
#define PVI_MAX_M 2 /* input dimensions: 1, 2 (discretizer.py:271-306) */
#define PVI_MAX_TRIG 4
#define PVI_MAX_OBS 8 /* axis-aligned obstacle boxes of a system's isavalidstate */

/* pvi_desc.flags */
#define PVI_FLAG_EXT_J_SLACK 1 /* the ext_J buffers have >= 64 readable bytes behind the stored rows (lets the 4-D
                                  window fill use 16-byte loads that may run past the end of a row) */

#define PVI_FLAG_HARD_INF 2    /* base-class semantics of DynamicProgramming.compute_backward_step (dynamicprogramming.py:
                                  195-236): an invalid action / next state costs exactly INF instead of the look-up-table
                                  class's INF + alpha*J_interp (:567).  The two differ only where isavalidstate rejects
                                  states INSIDE the grid box (obstacles): honoured by the n = 3 obstacle dynamics */

#define PVI_FLAG_F32_FEEDBACK 4 /* float32 handles on the 4-D window sweep (k_sweep_lean4): error-feedback storage of J.  A
                                  float32 J that grows by a near-constant increment per sweep is rounded the same way sweep
                                  after sweep; on BASELINE configs[2] the iterates drift to 1.5e-5 of max J around sweep
                                  1 500 before the contraction pulls them back (J* itself: 2e-6).  With this flag every node
                                  keeps the rounding residual of its own stored value (one more float per node, private to
                                  the node): the backup of the chosen action is re-evaluated in float64 (float64
                                  dynamics, the float32 window values), the residual of the previous sweep is added, the sum is stored as float32 and the
                                  new residual kept -- first-order noise shaping: roundings no longer add up over the
                                  sweeps.  What the gathers read stays one float32 per node.  pvi_create fails with
                                  PVI_EINVAL when the handle does not take that sweep (other dtype / dimension / a window
                                  that does not fit LDS); pvi_set_J and pvi_terminal_cost clear the residuals. */

/* error codes */
#define PVI_OK 0
#define PVI_EINVAL -1   /* bad argument / unsupported shape (reference: ValueError / NotImplementedError) */
#define PVI_EHIP -2     /* HIP runtime error, see pvi_last_error() */
#define PVI_ENOMEM -3
#define PVI_ESTATE -4   /* call not valid in the handle's current state */
#define PVI_EHALO -5    /* a gather left the stored slab: halo too small for this dynamics */
#define PVI_ECORRUPT -6 /* pvi_sweep: the corruption detector of the error-feedback sweep fired (pvi_override("FBCHECK", "1")): the
                           float32 action loop and the float64 epilogue disagree about the same backup beyond any rounding */

/* storage / interpolation arithmetic type of J on the device */
#define PVI_F32 0
#define PVI_F64 1

/* dynamics evaluated in-kernel (dx = f(x,u), pyro/dynamic/mechanical.py:238-263) */
#define PVI_DYN_TABLE 0     /* none: x_next / G tables supplied by the host (pvi_set_tables) */
#define PVI_DYN_PENDULUM 1  /* pyro/dynamic/pendulum.py:16 SinglePendulum, :283 InvertedPendulum */
#define PVI_DYN_CARTPOLE 2  /* pyro/dynamic/cartpole.py:322 CartPole; dyn_params = [m1+m2, m2 lcg, m2 lcg^2, -m2 lcg, m2 g lcg]
                               (five values; nothing beyond them is read) */
#define PVI_DYN_TWOLINK 3   /* pyro/dynamic/manipulator.py:795 TwoLinkManipulator, pendulum.py:340 DoublePendulum */
/* Any MechanicalSystem (mechanical.py:222-263: ddq = inv(H)(B u - C dq - g - d), x = [q; dq]) through per-node tables
   evaluated once by the host -- O(N) calls of the system's own H, C, B, g, d instead of the O(N*A) of the look-up
   tables: the acceleration is affine in u,  ddq = a0(q, dq) + Bn(q) u  with
     trig[0] = a0 = inv(H)(-C dq - g - d)   [all grid nodes, C order][dof]
     trig[1] = Bn = inv(H) B                [position nodes, C order][dof][m]
   (float64).  dyn_params is unused.  NODE_dxm: dof = d, m inputs. */
#define PVI_DYN_NODE_1x1 4  /* e.g. pyro/dynamic/mountaincar.py:22 MountainCar */
#define PVI_DYN_NODE_2x1 5  /* e.g. pendulum.py Acrobot (dof 2, one actuator) */
#define PVI_DYN_NODE_2x2 6

/* Three-dimensional systems of the reference's value-iteration demos (x_grid n = 3, discretizer.py:201-215), closed
   forms in the reference's operation order; trig[] = host tables over the levels of the axis named below.
   Their isavalidstate adds axis-aligned obstacle boxes to the state box (pvi_desc.obs_*). */
#define PVI_DYN_HELICOPTER 7 /* pyro/dynamic/drone.py:547 ConstantSpeedHelicopterTunnel, x = [dy, y, x], m = 1:
                                f = [(1/mass) u, x0, vx]; dyn_params = [1/mass, vx] */
#define PVI_DYN_KINCAR 8     /* pyro/dynamic/vehicle_steering.py:20 KinematicBicyleModel and subclasses (:717 car,
                                :973 car with obstacles), x = [x, y, theta], u = [v, beta]:
                                f = [u0 cos x2, u0 sin x2, u0 tan(u1) (1/length)]; trig[0] = cos, trig[1] = sin over the
                                levels of axis 2; act_aux[a] = u0 tan(u1) (1/length) per action (host NumPy) */
#define PVI_DYN_QUARTERCAR 9 /* pyro/dynamic/suspension.py:20 QuarterCarOnRoughTerrain, x = [dy, y, x], m = 1:
                                f = [(1/mass)(u - k (x1 - z) - b (x0 - dz)), x0, vx]; trig[0] = z(x2), trig[1] = dz(x2)
                                over the levels of axis 2 (the system's own ground profile);
                                dyn_params = [1/mass, k, b, vx] */

#define PVI_DYN_HOLONOMIC 10 /* vehicle_steering.py:201 HolonomicMobileRobot (+ :336 withObstacles), x = [x, y], u = [vx, vy]:
                                f = [u0, u1]; obstacles: obs_half = 0 (a point robot) */
#define PVI_DYN_LONGCAR 11   /* pyro/dynamic/vehicle_propulsion.py:23 LongitudinalFrontWheelDriveCarWithWheelSlipInput,
                                x = [x, v], u = [slip]: f = [v, (mu m g rr - fd) / (m (1 + mu ry))]; its isavalidinput also
                                rejects negative wheel loads (evaluated per cell in-kernel).  trig[0] = drag force fd over
                                the levels of axis 1; act_aux = [A][2] = {mu m g rr, m (1 + mu ry)} (host NumPy);
                                dyn_params = [m, ry, m g rr, m g rf] */

#define PVI_DYN_CARTPOLE_SW 12 /* OPT-IN, float32 and 4-D window sweep only: the same cart-pole with its generalised coordinates in
                                the order q = (theta, x), i.e. state (theta, x, dtheta, dx); dyn_params as PVI_DYN_CARTPOLE, trig[0..1]
                                = cos / sin over the levels of axis 0.  The caller permutes levels, box, Q, S, xbar and transposes J / pi
                                (pyro_amd/planning/permuted.py).  pvi_create refuses it with float64, on a handle that does not get the
                                window sweep, and for tables / rollouts / pvi_eval_f. */

/* cost evaluated in-kernel */
#define PVI_COST_TABLE 0      /* G supplied by the host */
#define PVI_COST_QUADRATIC 1  /* pyro/analysis/costfunction.py:101 QuadraticCostFunction */
#define PVI_COST_TIME 2       /* costfunction.py:287 TimeCostFunction: g = 1 (0 inside the target ball), h = 0;
                                 uses xbar, EPS, INF, ontarget_check; Q, R, S, ubar are ignored */
#define PVI_COST_REACHABILITY 4 /* costfunction.py:421 Reachability with the default target test: g = 0 on states the system's
                                  isavalidstate accepts else INF, h = 0 inside ||x - xbar|| < EPS else INF; uses xbar, EPS, INF */
#define PVI_COST_QUADRATIC_DOMAIN 3 /* costfunction.py:339 QuadraticCostFunctionWithDomainCheck: the quadratic g / h, INF on
                                 states the system's isavalidstate rejects (box + obstacles), 0 on target (applied last) */

typedef struct pvi_problem* pvi_handle;

/*
 * Problem description = the constructor arguments of
 *   GridDynamicSystem(sys, x_grid_dim, u_grid_dim, dt)        pyro/planning/discretizer.py:27
 *   DynamicProgramming(grid_sys, cost_function)               pyro/planning/dynamicprogramming.py:119
 * flattened to plain data.
 */
typedef struct pvi_desc {
    uint32_t struct_size;            /* sizeof(pvi_desc), ABI guard */
    int32_t n;                       /* sys.n */
    int32_t m;                       /* sys.m */
    int32_t x_dim[PVI_MAX_N];        /* x_grid_dim */
    int32_t u_dim[PVI_MAX_M];        /* u_grid_dim */
    const double* x_level[PVI_MAX_N];/* np.linspace arrays, discretizer.py:142 (x_dim[d] doubles each) */
    const double* u_level[PVI_MAX_M];/* discretizer.py:158 */
    double x_lb[PVI_MAX_N];          /* sys.x_lb / x_ub: isavalidstate box, pyro/dynamic/system.py:198 */
    double x_ub[PVI_MAX_N];
    double u_lb[PVI_MAX_M];          /* sys.u_lb / u_ub: isavalidinput box, system.py:208 */
    double u_ub[PVI_MAX_M];
    double dt;                       /* grid_sys.dt */
    int32_t dtype;                   /* PVI_F32 | PVI_F64 */
    int32_t dynamics_id;             /* PVI_DYN_* */
    double dyn_params[16];           /* derived constants, see pyro_amd/dynamic/<system>.device_dynamics() */
    const double* trig[PVI_MAX_TRIG];/* host-computed sin/cos tables over the angle levels (dynamics
                                        specific, see pyrovi.hip); NULL -> computed with libm */
    int32_t cost_id;                 /* PVI_COST_* */
    int32_t ontarget_check;          /* cf.ontarget_check, costfunction.py:137 */
    double Q[PVI_MAX_N * PVI_MAX_N]; /* row-major n x n */
    double R[PVI_MAX_M * PVI_MAX_M];
    double S[PVI_MAX_N * PVI_MAX_N];
    double xbar[PVI_MAX_N];
    double ubar[PVI_MAX_M];
    double EPS;                      /* cf.EPS */
    double INF;                      /* cf.INF */
    /* slab of axis 0 handled by this handle (multi-GPU: one handle per rank); whole grid: 0..x_dim[0] */
    int32_t row_begin, row_end;      /* rows updated by a sweep */
    int32_t halo_lo, halo_hi;        /* extra rows of J kept below / above for the gathers */
    int32_t device;                  /* HIP device ordinal */
    int32_t flags;                   /* PVI_FLAG_* */
    /* optional caller-owned device buffers (e.g. torch tensors used for the RCCL halo exchange):
       two J buffers of stored_rows*plane elements of `dtype`, one pi buffer of owned_rows*plane
       bytes (A<=256) or uint16 (A<=65536).  NULL -> allocated by the library. */
    void* ext_J[2];
    void* ext_pi;
    /* ---- ABI 2 ---- */
    /* isavalidstate beyond the box (drone.py:590-611, vehicle_steering.py:1004-1021): x is invalid when, for some box b,
         x[obs_axis[0]] + obs_half[0] > obs_box[b][0]  and  x[obs_axis[1]] + obs_half[1] > obs_box[b][1]  and
         x[obs_axis[0]] - obs_half[0] < obs_box[b][2]  and  x[obs_axis[1]] - obs_half[1] < obs_box[b][3].
       Only read for the explicit dynamics PVI_DYN_HELICOPTER .. PVI_DYN_LONGCAR. */
    int32_t n_obs;
    int32_t obs_axis[2];
    double obs_half[2];
    double obs_box[PVI_MAX_OBS][4];
    const double* act_aux;           /* per-action constants of the dynamics (PVI_DYN_KINCAR: [A], PVI_DYN_LONGCAR: [A][2]) or NULL */
} pvi_desc;

/* ---- library ------------------------------------------------------------------------------ */
int pvi_abi_version(void);
const char* pvi_last_error(void);
int pvi_device_count(int* count);

/* Pin one of the kernel variants pvi_create chooses between by heuristics and timed sweeps (tile shapes, lanes per
   node, wave mappings, dense / sparse walks, which float32 path ...).  Every variant computes the same recursion: bit for
   bit within a dtype path, within the float32 tolerance across float32 paths -- this is how the tests compare them and
   how profiling passes keep set-up's timed candidates out of their averages.  Process-wide, read when a handle is
   created.  value == NULL removes the key, key == NULL removes all.  Unknown keys: PVI_EINVAL (the list is in
   pyro_amd/csrc/pyrovi.hip OVERRIDE_KEYS).  The library never reads the environment for any of this: no environment
   variable changes what it computes (PVI_RCCL_LIB, the file name of the RCCL library, is the one variable it reads). */
int pvi_override(const char* key, const char* value);

/* ---- problem life cycle ------------------------------------------------------------------- */
/* GridDynamicSystem.__init__ + DynamicProgramming.__init__ (discretizer.py:27, dynamicprogramming.py:119) */
int pvi_create(const pvi_desc* desc, pvi_handle* out);
void pvi_destroy(pvi_handle h);

/* geometry helpers */
int64_t pvi_plane_size(pvi_handle h);   /* nodes per axis-0 row */
int64_t pvi_stored_nodes(pvi_handle h); /* (row_end+halo_hi - (row_begin-halo_lo)) * plane */
int64_t pvi_owned_nodes(pvi_handle h);
int pvi_pi_itemsize(pvi_handle h);      /* bytes per policy entry on the device */
/* one-line description of the kernel path chosen for this problem (diagnostics / bench output) */
int pvi_describe(pvi_handle h, char* buf, int32_t n);

/* ---- cost-to-go ---------------------------------------------------------------------------- */
/* evaluate_terminal_cost: J[s] = cf.h(x_s), pi = 0 on all stored rows (dynamicprogramming.py:159-171) */
int pvi_terminal_cost(pvi_handle h);
/* upload / download rows [row0, row0+nrows) of the CURRENT cost-to-go (host float64, converted) */
int pvi_set_J(pvi_handle h, const double* J_rows, int32_t row0, int32_t nrows);
int pvi_get_J(pvi_handle h, double* J_rows, int32_t row0, int32_t nrows);
/* the cost-to-go of the previous sweep (dp.J_next, dynamicprogramming.py:181) */
int pvi_get_J_prev(pvi_handle h, double* J_rows, int32_t row0, int32_t nrows);
/* policy of the last sweep as int64 action ids, owned rows only (dynamicprogramming.py:570) */
int pvi_get_pi(pvi_handle h, int64_t* pi_rows, int32_t row0, int32_t nrows);

/* ---- the hot path --------------------------------------------------------------------------- */
/*
 * initialize_backward_step + compute_backward_step + finalize_backward_step, `max_sweeps` times
 * (dynamicprogramming.py:175-261, :557-570; drivers :265-314).  One fused kernel per sweep:
 *   x_next = f(x,u)*dt + x ; validity ; G = g*dt | INF ; J_interp(x_next) ; min / first argmin.
 * tol >= 0: stop after the first sweep whose delta = max(|dmax|,|dmin|) <= tol (the stop is
 * decided on the device, later launches of the batch turn into no-ops).  tol < 0: never stop.
 * stats: [max_sweeps][4] doubles = (max J, max(J-J_prev), min(J-J_prev), delta) per executed sweep.
 * Only valid when the handle stores the rows its gathers need (single GPU: whole grid).
 */
int pvi_sweep(pvi_handle h, int32_t max_sweeps, double alpha, double tol, double* stats, int32_t* sweeps_done);
/* GPU time of the sweep kernels of the last pvi_sweep call, from HIP events on the handle's stream */
int pvi_last_sweep_ms(pvi_handle h, float* ms);

/* multi-GPU building blocks: one sweep of the owned rows enqueued on `stream` (NULL: the
   handle's stream), no host synchronisation; the caller exchanges halo rows of the new current
   buffer (pvi_device_J) before the next call and reduces the stats across ranks. */
int pvi_sweep_async(pvi_handle h, double alpha, void* stream);
int pvi_sweep_stats(pvi_handle h, double stats3[3], void* stream); /* synchronises `stream` */
int pvi_device_J(pvi_handle h, int which /*0 current, 1 previous*/, void** dev_ptr);
int pvi_device_pi(pvi_handle h, void** dev_ptr);
int pvi_synchronize(pvi_handle h);

/* diagnostics (sanitizer-style cross check): one backup of the CURRENT cost-to-go computed twice -- by the handle's
   production kernel path (LDS windows, set-up tables, float32 displacement, ...) and by the plain-gather kernel with
   float64 dynamics -- and compared on the device: max |J_a - J_b| / max |J_b| and the number of nodes whose action
   differs.  The current cost-to-go is left alone; the previous-sweep buffer and the policy are overwritten as by a sweep
   that is not counted.  float64 handles: both paths are exact, the result must be 0 / 0. */
int pvi_self_check(pvi_handle h, double alpha, double* max_rel_diff, int64_t* pi_mismatches);

/* diagnostics, host only (no device is touched; callable on a machine without a GPU): the tiling of one velocity plane that
   the 4-D float32 sweep would use.  corner0[V0] = axis-0 corner index of the position row reached from
   velocity row j (negative: the row leaves the box) -- tiles never straddle a change of it; `cap` = most rows per tile,
   `threads` = workgroup size (a multiple of 64, <= 512), `wmax` = widest tile.  Writes up to `max_tiles` rectangles
   {row0, nrows, col0, ncols} and returns their number (or a negative error code).  The rectangles partition the V0 x V1
   plane and hold at most `threads` nodes each. */
int pvi_plan_plane_tiles(int32_t V0, int32_t V1, const int32_t* corner0, int32_t cap, int32_t threads, int32_t wmax,
                         int32_t* tiles4, int32_t max_tiles);
/* ... and the launch order of `rows` x `n1` position nodes x `tiles_per_plane` tiles in `bands` bands: out[k] = tile id
   ((r * n1 + i1) * tiles_per_plane + t) of physical block k, 0xffffffff = padding; block k runs on XCD k % 8, XCD x sweeps
   its chunk of axis 1 for every row in turn.  Returns the number of blocks (a multiple of 8), writing at most `max_blocks`. */
int64_t pvi_plan_schedule(int32_t rows, int32_t n1, int32_t tiles_per_plane, int32_t bands, uint32_t* out, int64_t max_blocks);
/* ... with tile lists of different lengths per row of axis 0 (a corner step that falls on a level lands a row earlier or later):
   counts[r] <= tiles_per_plane = the tiles row r really has; ids t >= counts[r] are NOT scheduled (rounds 3-4 launched a
   workgroup for each of them: 8.4 % of the workgroups of BASELINE configs[2]).  counts == NULL: every row has tiles_per_plane. */
int64_t pvi_plan_schedule_rows(int32_t rows, int32_t n1, int32_t tiles_per_plane, int32_t bands, const int32_t* counts, uint32_t* out,
                               int64_t max_blocks);

/* ---- tables (tier B and reference attributes) ---------------------------------------------- */
/* compute_xnext_table / compute_action_set_table / compute_cost_lookuptable for rows
   [row0,row0+nrows) (discretizer.py:342-376, :314-338; dynamicprogramming.py:517-553).
   Any output pointer may be NULL.  Shapes: x_next [nodes][A][n], masks [nodes][A], G [nodes][A]. */
int pvi_build_tables(pvi_handle h, int32_t row0, int32_t nrows, double* x_next, uint8_t* x_next_isok,
                     uint8_t* action_isok, double* G);
/* tier B: host-built tables for arbitrary sys.f / cf.g (dynamics_id == PVI_DYN_TABLE, desc.INF = cf.INF).
   ok == NULL : look-up-table semantics, Q = G + alpha*J_interp(x_next) with G already INF on invalid cells
                (DynamicProgrammingWithLookUpTable, dynamicprogramming.py:564-570);
   ok != NULL : ok[node][a] = action_isok & x_next_isok; invalid cells cost exactly INF
                (base class DynamicProgramming.compute_backward_step, dynamicprogramming.py:195-236). */
int pvi_set_tables(pvi_handle h, const double* x_next, const double* G, const uint8_t* ok);

/* PolicyEvaluator.__init__ / compute_lookuptable (dynamicprogramming.py:623-677, :704-735): the one-action-per-node tables
   of a GIVEN control law on the whole grid,
     u = ctl.c(x, rbar, t) ; x_next = f(x, u) dt + x ; ok = isavalidinput(x, u) and isavalidstate(x_next) ; G = g(x, u) dt | INF
   on a handle with closed-form mechanical dynamics and the in-kernel quadratic cost (else PVI_ESTATE).
     PVI_CTL_TABLE            U [nodes][m] is supplied by the caller (any Python controller, evaluated on the host)
     PVI_CTL_COMPUTED_TORQUE  pyro/control/nonlinear.py:23-116 on a fully actuated system (pendulum family, two-link arm);
                              ctl_params = [q_d[dof], 2 zeta w0, w0^2]; U is an output
   Outputs (host): x_next [nodes][n], ok [nodes], G [nodes]; feed them to a PVI_DYN_TABLE handle with one action. */
#define PVI_CTL_TABLE 0
#define PVI_CTL_COMPUTED_TORQUE 1
int pvi_policy_tables(pvi_handle h, int32_t controller_id, const double* ctl_params, double* U, double* x_next, uint8_t* ok,
                      double* G);

/* ---- interpolant of J_k ------------------------------------------------------------------------------ */
#define PVI_INTERP_LINEAR 0          /* RegularGridInterpolator('linear', fill 0), discretizer.py:570-587 (default) */
#define PVI_INTERP_BICUBIC_SPLINE 1  /* RectBivariateSpline(kx=ky=3, s=0), discretizer.py:590-612: refit every sweep,
                                        x_next clamped to the grid box, no zero fill */
#define PVI_INTERP_NEAREST 2         /* RegularGridInterpolator('nearest', fill 0): per axis the nearer level, ties to the lower one
                                        (scipy _evaluate_nearest).  Table tier (PVI_DYN_TABLE) only; before pvi_set_tables. */
/* PVI_INTERP_BICUBIC_SPLINE turns the handle into DynamicProgramming2DRectBivariateSpline
   (dynamicprogramming.py:578-614): n == 2, whole-grid handles only, >= 4 levels per axis; works with the
   pendulum-family in-kernel dynamics (look-up-table semantics) and with tier-B tables.  On a tier-B handle call it
   BEFORE pvi_set_tables: the linear sweep keeps only packed records of the tables, the spline sweep the raw tables
   (PVI_ESTATE otherwise). */
int pvi_set_interpolation(pvi_handle h, int32_t kind);
/* fit the spline through the CURRENT cost-to-go and return its B-spline coefficients [x_dim0][x_dim1]
   (RectBivariateSpline.get_coeffs() of dp.J_interpol, dynamicprogramming.py:594) */
int pvi_spline_coefficients(pvi_handle h, double* coef);

/* ---- policy consumers (SURVEY 8f "next": the step after the path) ------------------------------------ */
/* replace the device policy, e.g. after clean_infeasible_set on the host (dynamicprogramming.py:322-334) */
int pvi_set_pi(pvi_handle h, const int64_t* pi_rows, int32_t row0, int32_t nrows);
/* B closed-loop Euler rollouts of the look-up-table policy: u = LookUpTableController.c(x) (n-linear
   interpolation of the INPUT values selected by pi, 0 outside the grid; dynamicprogramming.py:72-107),
   dx = f(x,u) (pyro/control/controller.py:328-355), x_{i+1} = dx*dt + x_i (pyro/analysis/simulation.py:298-324).
   X_traj [B][npts][n] and U_traj [B][npts][m] may be NULL; X_end [B][n] may be NULL.  float64, whole-grid handle
   with closed-form in-kernel dynamics: the mechanical closed forms and the explicit systems PVI_DYN_HELICOPTER ..
   PVI_DYN_LONGCAR (per-node-table dynamics PVI_DYN_NODE_* and table handles only know f on the grid: PVI_ESTATE). */
int pvi_rollout(pvi_handle h, int64_t B, const double* X0, int32_t npts, double dt, double* X_traj, double* U_traj,
                double* X_end);
/* The sweeps need f only at grid nodes and grid actions (host tables over the levels); a rollout needs it at arbitrary
   (x, u).  Constants of the CONTINUOUS closed form that are not in dyn_params (n <= 64 doubles), required before
   pvi_rollout for:
     PVI_DYN_QUARTERCAR  [terms, a[terms], w[terms], phi[terms]]: ground z = sum a sin(w (x - phi))   suspension.py:73-95
     PVI_DYN_LONGCAR     [mu_max, mu_slope, rho cdA, m, g, ry, rr]                                     vehicle_propulsion.py:96-184
   (helicopter, kinematic car, point robot and the mechanical closed forms need none). */
int pvi_set_rollout_params(pvi_handle h, const double* params, int32_t n);

/* ---- multi-GPU: axis-0 slabs, one process per GPU, RCCL inside the library (no reference counterpart) ------ */
/* Rank r owns a contiguous block of rows of axis 0 and stores `halo_rows` more on either side.  One sweep =
   boundary kernels -> [halo exchange with the +-1 neighbours (ncclSend / ncclRecv in one group) || interior kernel];
   the statistics (max J, max d, min d) are all-reduced (ncclAllReduce max, 3 doubles) for the stop test / the last
   sweep.  Slabs thinner than the halo fall back to broadcasting every slab.  Results are those of pvi_sweep on the
   whole grid, bit for bit.  RCCL is loaded at run time (librccl.so.1; PVI_RCCL_LIB overrides the name). */
#define PVI_COMM_ID_BYTES 128
typedef struct pvi_shard_s* pvi_shard;
/* ncclGetUniqueId: call on ONE rank and hand the 128 bytes to the others by any means (MPI, a file, a TCP store) */
int pvi_comm_unique_id(uint8_t* id128);
/* `whole` describes the WHOLE grid (row_begin / row_end / halo_* / ext_* are ignored; device = this rank's GPU).
   halo_rows = ceil(max |x_next_0 - x_0| / dx_0) + 1 (mechanical systems: max |dq_0| dt / dx_0) and must be ONE number for
   the whole grid: the counts of the send / recv pairs and the send/recv-or-broadcast decision depend on it.
     halo_rows >= 1 : the width, the same on every rank -- checked across the ranks, PVI_EINVAL if they disagree;
     halo_rows <  0 : -halo_rows is a bound THIS rank derived from its own rows (e.g. from its part of an x_next
                      table); the library takes the largest over the ranks.
   id128 may be NULL when world == 1 (no communicator).  overlap != 0: boundary-first schedule on two streams.
   Collective: every rank of the communicator must call it (the ranks agree on the halo and on success; if one rank
   cannot build its slab, all return an error). */
int pvi_shard_create(const pvi_desc* whole, int32_t rank, int32_t world, int32_t halo_rows, const uint8_t* id128,
                     int32_t overlap, pvi_shard* out);
/* Bring-your-own transport (MPI without RCCL, or a host-staged test harness): the same slab schedule with the two
   inter-rank steps handed to the caller.  Both callbacks run on the calling host thread, synchronously.
     sendrecv : exchange halo rows of the current cost-to-go with the +-1 neighbours.  The four buffers are DEVICE
                pointers (NULL where there is no neighbour); the data to send is complete once `stream` has drained, so
                a host-staged implementation synchronises `stream` first; received rows must be in place on return.
     max3     : in-place maximum over all ranks of three host doubles.
   Slabs must be at least `halo_rows` thick (no broadcast fall-back on this path). */
typedef struct pvi_transport {
    void* user;
    int (*sendrecv)(void* user, const void* send_lo, void* recv_lo, size_t lo_send_bytes, size_t lo_recv_bytes,
                    const void* send_hi, void* recv_hi, size_t hi_send_bytes, size_t hi_recv_bytes, void* stream);
    int (*max3)(void* user, double v[3]);
} pvi_transport;
int pvi_shard_create_with_transport(const pvi_desc* whole, int32_t rank, int32_t world, int32_t halo_rows,
                                    const pvi_transport* transport, int32_t overlap, pvi_shard* out);
void pvi_shard_destroy(pvi_shard s);
int pvi_shard_rows(pvi_shard s, int32_t* row_begin, int32_t* row_end);
int pvi_shard_terminal_cost(pvi_shard s);
/* compute_steps / solve_bellman_equation on the sharded grid (dynamicprogramming.py:265-314): up to max_sweeps backups,
   stop when tol >= 0 and delta <= tol (every rank stops at the same sweep: the statistics are all-reduced).
   stats4 = (max J, max d, min d, delta) of the WHOLE grid for the last executed sweep. */
int pvi_shard_sweep(pvi_shard s, int32_t max_sweeps, double alpha, double tol, double* stats4, int32_t* sweeps_done);
/* tier B on a slab (dynamics_id == PVI_DYN_TABLE): the look-up tables and the initial cost-to-go of THIS rank's rows;
   pvi_shard_set_J also exchanges the halos (call it on every rank) */
int pvi_shard_set_tables(pvi_shard s, const double* x_next_rows, const double* G_rows, const uint8_t* ok_rows);
int pvi_shard_set_J(pvi_shard s, const double* J_owned_rows);
int pvi_shard_get_J(pvi_shard s, double* J_owned_rows);   /* this rank's rows, float64 */
int pvi_shard_get_J_prev(pvi_shard s, double* J_owned_rows); /* ... of the previous sweep (dp.J_next) */
int pvi_shard_halo(pvi_shard s, int32_t* halo_rows);      /* the width the ranks agreed on (rows stored beyond the owned ones) */
int pvi_shard_get_pi(pvi_shard s, int64_t* pi_owned_rows);
/* ---- ABI 3: the sharded solve behind the reference's attributes ---- */
/* dp.J (which = 0) / dp.J_next (which = 1: the cost-to-go of the previous sweep) and dp.pi of the WHOLE grid on every
   rank (dynamicprogramming.py:181, :569-570).  Collective over the RCCL communicator (every rank calls; one ncclBroadcast
   per owner in one group).  With a caller-supplied transport: PVI_ESTATE for world > 1 -- gather the rows of
   pvi_shard_get_J / pvi_shard_get_pi with that transport. */
int pvi_shard_gather_J(pvi_shard s, int32_t which, double* J_whole);
int pvi_shard_gather_pi(pvi_shard s, int64_t* pi_whole);
/* finalize_backward_step reports every sweep (dynamicprogramming.py:240-261): on != 0 makes a fixed-count
   pvi_shard_sweep take and all-reduce the statistics after EVERY sweep (a host synchronisation each; off by default).
   pvi_shard_sweep_history returns the rows (max J, max d, min d, delta) of the last call's sweeps that took
   statistics, oldest first: all of them with a stop tolerance or the switch on, else the last sweep only. */
int pvi_shard_stats_every_sweep(pvi_shard s, int32_t on);
int pvi_shard_sweep_history(pvi_shard s, double* stats, int32_t max_rows, int32_t* rows);
/* "rank=r/w rows=[..) stored=[..) halo=h exchange=send/recv|broadcast[+overlap] pieces=k comm=rccl|caller|none
   rccl_ranks=N | <pvi_describe of the interior handle>"; rccl_ranks = ncclCommCount of the communicator in use */
int pvi_shard_describe(pvi_shard s, char* buf, int32_t n);
/* GPU times of the last pvi_shard_sweep call on this rank (HIP events on its compute and comm streams), per-sweep
   averages over its last <= 32 sweeps, milliseconds: out6 = {boundary kernels, interior kernel, halo exchange,
   exchange time NOT hidden behind the interior kernel, boundary + interior + exposed exchange, sweeps averaged over} */
int pvi_shard_timing(pvi_shard s, double out6[6]);

/* ---- batched dynamics ------------------------------------------------------------------------ */
/* dX[b] = f(X[b], U[b]) for B states (mechanical.py:238-263); host pointers, float64.  dyn_params points to 16 doubles (as
   pvi_desc.dyn_params), all of them copied to the device (pad the tail with zeros); PENDULUM, CARTPOLE, TWOLINK only. */
int pvi_eval_f(int32_t dynamics_id, const double* dyn_params, int32_t n, int32_t m, int64_t B, const double* X,
               const double* U, double* dX);

#ifdef __cplusplus
}
#endif
#endif /* PYROVI_H */
