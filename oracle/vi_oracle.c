/*
 * CPU ORACLE (test infrastructure only) -- C/OpenMP twin of oracle/vi_oracle.py.
 *
 * Same arithmetic, operation for operation (build with -ffp-contract=off): float64, the
 * reference's operation order, scipy-equivalent n-linear interpolation.  It exists so that
 * (i) full-size configurations can be checked on the GPU box in seconds and (ii) bench.py has a
 * multi-core CPU baseline ("port") to time beside the HIP kernels.  Pinned: tests/test_c_oracle.py
 * requires bit-identical results to the NumPy oracle, which is itself pinned to the reference's
 * golden vectors.  Never linked or imported by the product.
 *
 * Reference lines restated (relative to the reference repo root):
 *   x_next = f(x,u)*dt + x, validity          pyro/planning/discretizer.py:342-376, pyro/dynamic/system.py:198-215
 *   ddq = inv(H)(B u - C dq - g - d)          pyro/dynamic/mechanical.py:222-234 (+ pendulum.py, cartpole.py, manipulator.py)
 *   G = g*dt | INF                            pyro/planning/dynamicprogramming.py:517-553, pyro/analysis/costfunction.py:169-204
 *   Q = G + alpha*J_interp(x_next); min/argmin pyro/planning/dynamicprogramming.py:557-570
 */
#include <math.h>
#include <stdint.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

#define DYN_PENDULUM 1
#define DYN_CARTPOLE 2
#define DYN_TWOLINK 3

typedef struct {
    int32_t n, m, A, dyn;
    int32_t dim[4];
    const double* lev[4];
    const double* trig[4]; /* same tables the GPU consumes */
    const double* utab;    /* [A][m] */
    double lb[4], ub[4];   /* isavalidstate box */
    double u_lb[2], u_ub[2];
    double dt;
    double c[16];
    double Q[16], R[4], S[16], xbar[4], ubar[2];
    double EPS, INF;
    int32_t ontarget;
} vio_problem;

static double quad_form(const double* M, const double* dx, int n) {
    double out = 0.0;
    for (int i = 0; i < n; ++i) {
        double row = M[i * n] * dx[0];
        for (int j = 1; j < n; ++j) row = row + M[i * n + j] * dx[j];
        double term = dx[i] * row;
        out = (i == 0) ? term : out + term;
    }
    return out;
}

static double l2norm(const double* dx, int n) {
    double s = dx[0] * dx[0];
    for (int j = 1; j < n; ++j) s = s + dx[j] * dx[j];
    return sqrt(s);
}

/* np.searchsorted(grid, x, 'right') - 1 clipped to [0, N-2] */
static int find_interval(const double* g, int N, double x) {
    int lo = 0, hi = N; /* first index with g[idx] > x */
    while (lo < hi) {
        int mid = (lo + hi) >> 1;
        if (g[mid] <= x) lo = mid + 1; else hi = mid;
    }
    int i = lo - 1;
    if (i < 0) i = 0;
    if (i > N - 2) i = N - 2;
    return i;
}

static void accel(const vio_problem* P, const double* x, const double* tr, const double* u, double* a) {
    const double* c = P->c;
    if (P->dyn == DYN_PENDULUM) {
        double g = c[1] * tr[0], d = c[2] * x[1];
        double rhs = (u[0] - g) - d;
        a[0] = c[0] * rhs;
    } else if (P->dyn == DYN_CARTPOLE) {
        double cth = tr[0], sth = tr[1], dth = x[3];
        double H00 = c[0], H01 = c[1] * cth, H11 = c[2];
        double C01 = (c[3] * sth) * dth;
        double Cdq0 = C01 * dth;
        double g1 = c[4] * sth;
        double r0 = u[0] - Cdq0, r1 = -g1;
        double det = H00 * H11 - H01 * H01;
        double i00 = H11 / det, i01 = -H01 / det, i10 = -H01 / det, i11 = H00 / det;
        a[0] = i00 * r0 + i01 * r1;
        a[1] = i10 * r0 + i11 * r1;
    } else {
        double s1 = tr[0], c2 = tr[1], s2 = tr[2], s12 = tr[3];
        double dq0 = x[2], dq1 = x[3];
        double H00 = (c[0] + c[1] * (c[2] + c[3] * c2)) + c[4];
        double H01 = (c[5] + c[6] * c2) + c[4];
        double H11 = c[5] + c[4];
        double h = c[6] * s2;
        double C00 = -h * dq1, C10 = h * dq0, C01 = -h * (dq0 + dq1);
        double Cdq0 = C00 * dq0 + C01 * dq1, Cdq1 = C10 * dq0;
        double G0 = -c[7] * s1 - c[8] * s12, G1 = -c[8] * s12;
        double D0 = c[9] * dq0, D1 = c[10] * dq1;
        double r0 = ((u[0] - Cdq0) - G0) - D0;
        double r1 = ((u[1] - Cdq1) - G1) - D1;
        double det = H00 * H11 - H01 * H01;
        double i00 = H11 / det, i01 = -H01 / det, i10 = -H01 / det, i11 = H00 / det;
        a[0] = i00 * r0 + i01 * r1;
        a[1] = i10 * r0 + i11 * r1;
    }
}

static void state_trig(const vio_problem* P, const int* idx, double* tr) {
    if (P->dyn == DYN_PENDULUM) {
        tr[0] = P->trig[0][idx[0]];
    } else if (P->dyn == DYN_CARTPOLE) {
        tr[0] = P->trig[0][idx[1]];
        tr[1] = P->trig[1][idx[1]];
    } else {
        tr[0] = P->trig[0][idx[0]];
        tr[1] = P->trig[1][idx[1]];
        tr[2] = P->trig[2][idx[1]];
        tr[3] = P->trig[3][(int64_t)idx[0] * P->dim[1] + idx[1]];
    }
}

static double interp(const vio_problem* P, const double* J, const int64_t* strd, const double* xn) {
    int n = P->n, ci[4];
    double y[4];
    int oob = 0;
    for (int d = 0; d < n; ++d) {
        const double* g = P->lev[d];
        int N = P->dim[d];
        ci[d] = find_interval(g, N, xn[d]);
        y[d] = (xn[d] - g[ci[d]]) / (g[ci[d] + 1] - g[ci[d]]);
        oob |= (xn[d] < g[0]) | (xn[d] > g[N - 1]);
    }
    if (oob) return 0.0;
    int64_t base = 0;
    for (int d = 0; d < n; ++d) base += ci[d] * strd[d];
    if (n == 2) {
        double v00 = J[base], v01 = J[base + strd[1]], v10 = J[base + strd[0]], v11 = J[base + strd[0] + strd[1]];
        double a0 = 1.0 - y[0], a1 = 1.0 - y[1];
        return v00 * a0 * a1 + v01 * a0 * y[1] + v10 * y[0] * a1 + v11 * y[0] * y[1];
    }
    double val = 0.0;
    for (int corner = 0; corner < (1 << n); ++corner) {
        double w = 1.0;
        int64_t off = base;
        for (int d = 0; d < n; ++d) {
            int bit = (corner >> (n - 1 - d)) & 1;
            w = w * (bit ? y[d] : (1.0 - y[d]));
            if (bit) off += strd[d];
        }
        val = val + J[off] * w;
    }
    return val;
}

/* per-node prologue shared by the sweep and the Q probes */
typedef struct {
    double x[4], tr[4], gx;
    int on_target;
} node_ctx;

static void node_prologue(const vio_problem* P, int64_t node, node_ctx* c) {
    const int n = P->n;
    int idx[4];
    int64_t rem = node;
    for (int d = n - 1; d >= 0; --d) { idx[d] = (int)(rem % P->dim[d]); rem /= P->dim[d]; }
    double dx[4];
    for (int d = 0; d < n; ++d) { c->x[d] = P->lev[d][idx[d]]; dx[d] = c->x[d] - P->xbar[d]; }
    state_trig(P, idx, c->tr);
    c->gx = quad_form(P->Q, dx, n);
    c->on_target = P->ontarget && (l2norm(dx, n) < P->EPS);
}

/* Q[s,a] = G + alpha * J_interp(x_next)   (dynamicprogramming.py:534-549, :567) */
static double cell_q(const vio_problem* P, const int64_t* strd, const double* Jin, const node_ctx* c, int a,
                     double alpha) {
    const int n = P->n, m = P->m, dof = n / 2;
    const double* x = c->x;
    double u[2], du[2], acc[2], xn[4];
    int aok = 1;
    for (int k = 0; k < m; ++k) {
        u[k] = P->utab[a * m + k];
        du[k] = u[k] - P->ubar[k];
        aok &= !(u[k] < P->u_lb[k]) & !(u[k] > P->u_ub[k]);
    }
    accel(P, x, c->tr, u, acc);
    int ok = aok;
    for (int i = 0; i < dof; ++i) {
        xn[i] = x[dof + i] * P->dt + x[i];
        xn[dof + i] = acc[i] * P->dt + x[dof + i];
    }
    for (int d = 0; d < n; ++d) ok &= !(xn[d] < P->lb[d]) & !(xn[d] > P->ub[d]);
    double g = c->on_target ? 0.0 : (c->gx + quad_form(P->R, du, m));
    double G = ok ? g * P->dt : P->INF;
    return G + alpha * interp(P, Jin, strd, xn);
}

static void grid_strides(const vio_problem* P, int64_t* strd) {
    int64_t s = 1;
    for (int d = P->n - 1; d >= 0; --d) { strd[d] = s; s *= P->dim[d]; }
}

static void backup_node(const vio_problem* P, const int64_t* strd, const double* Jin, double* Jout, int64_t* pi,
                        double alpha, int64_t node, int32_t f32_storage) {
    node_ctx c;
    node_prologue(P, node, &c);
    double best = 0.0;
    int64_t arg = 0;
    for (int a = 0; a < P->A; ++a) {
        double q = cell_q(P, strd, Jin, &c, a, alpha);
        if (a == 0 || q < best) { best = q; arg = a; }  /* first minimum, np.argmin (:569-570) */
    }
    Jout[node] = f32_storage ? (double)(float)best : best;
    if (pi) pi[node] = arg;
}

/* one Bellman backup of nodes [node0, node1); J is the full grid; f32_storage rounds J_out to float */
void vio_sweep(const vio_problem* P, const double* Jin, double* Jout, int64_t* pi, double alpha, int64_t node0,
               int64_t node1, int32_t f32_storage, int32_t nthreads) {
    int64_t strd[4];
    grid_strides(P, strd);
#ifdef _OPENMP
    if (nthreads > 0) omp_set_num_threads(nthreads);
#endif
    /* (blocks of 512 nodes dealt on demand: in-box cells cost several times an out-of-box cell, and those come in
       long contiguous runs of the C-order node numbering) */
#pragma omp parallel for schedule(dynamic, 512)
    for (int64_t node = node0; node < node1; ++node) backup_node(P, strd, Jin, Jout, pi, alpha, node, f32_storage);
}

/* `nsweeps` whole-grid backups inside ONE parallel region (bench.py's cpu_baseline): the two caller-owned buffers
   ping-pong, no allocation and no thread start-up between sweeps.  The result is in J[nsweeps & 1]. */
void vio_sweeps(const vio_problem* P, double* J0, double* J1, int64_t* pi, double alpha, int32_t nsweeps,
                int32_t f32_storage, int32_t nthreads) {
    int64_t strd[4], N = 1;
    grid_strides(P, strd);
    for (int d = 0; d < P->n; ++d) N *= P->dim[d];
#ifdef _OPENMP
    if (nthreads > 0) omp_set_num_threads(nthreads);
#endif
#pragma omp parallel
    {
        for (int k = 0; k < nsweeps; ++k) {
            const double* Jin = (k & 1) ? J1 : J0;
            double* Jout = (k & 1) ? J0 : J1;
#pragma omp for schedule(dynamic, 512)
            for (int64_t node = 0; node < N; ++node) backup_node(P, strd, Jin, Jout, pi, alpha, node, f32_storage);
            /* (implicit barrier: the next sweep reads what every thread wrote) */
        }
    }
}

/* Q of chosen (node, action) pairs and the minimum over the actions of the same nodes: the tests' policy check
   (regret of a policy = Q[s, pi_test[s]] - min_a Q[s, a]) at sizes the NumPy oracle is slow at */
void vio_q_at(const vio_problem* P, const double* Jin, const int64_t* nodes, const int64_t* actions, int64_t count,
              double alpha, double* q_out, double* qmin_out) {
    int64_t strd[4];
    grid_strides(P, strd);
#pragma omp parallel for schedule(static)
    for (int64_t i = 0; i < count; ++i) {
        node_ctx c;
        node_prologue(P, nodes[i], &c);
        q_out[i] = cell_q(P, strd, Jin, &c, (int)actions[i], alpha);
        double best = 0.0;
        for (int a = 0; a < P->A; ++a) {
            double q = cell_q(P, strd, Jin, &c, a, alpha);
            if (a == 0 || q < best) best = q;
        }
        qmin_out[i] = best;
    }
}

/* J0 = h(x) (dynamicprogramming.py:159-171) */
void vio_terminal_cost(const vio_problem* P, double* J) {
    const int n = P->n;
    int64_t N = 1;
    for (int d = 0; d < n; ++d) N *= P->dim[d];
#pragma omp parallel for schedule(static)
    for (int64_t node = 0; node < N; ++node) {
        int64_t rem = node;
        double dx[4];
        for (int d = n - 1; d >= 0; --d) { dx[d] = P->lev[d][rem % P->dim[d]] - P->xbar[d]; rem /= P->dim[d]; }
        double h = quad_form(P->S, dx, n);
        if (P->ontarget && l2norm(dx, n) < P->EPS) h = 0.0;
        J[node] = h;
    }
}

int vio_max_threads(void) {
#ifdef _OPENMP
    return omp_get_max_threads();
#else
    return 1;
#endif
}
