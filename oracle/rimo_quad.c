/* rimo_quad.c -- oracle restatement of the GSL routines the hot path calls
 * through src/gsl.rs (TEST INFRASTRUCTURE ONLY, see rimo.h).
 *
 *   gsl_integration_qag, key 3 (QUADPACK dqage + dqk31)   <- gsl.rs:168-181
 *   gsl_deriv_central                                     <- gsl.rs:233-257
 *   gsl_sf_hyperg_2F1(a,b,c,1)                            <- gsl.rs:261-263
 *
 * GSL itself is a third-party system library absent from /root/reference and
 * from this image (gsl-sys/build.rs:11-14 accepts any version >= 1.0), so what
 * follows restates the published QUADPACK/GSL 2.x algorithms: qk.c (rule
 * application + rescale_error), qag.c (bisection loop, round-off and
 * singularity detectors), qpsrt.c/util.c (descending error list), deriv.c.
 */
#include <stdio.h>
#include <stdlib.h>
#include "rimo.h"
#include "rimo_math.h"
#include "../rimphony_amd/csrc/gk31_table.h"

static const double GK_X[32] = RIM_GK31_X;
static const double GK_WK[32] = RIM_GK31_WK;
static const double GK_WG[32] = RIM_GK31_WG;

/* qk.c: rescale_error */
static double rescale_error(double err, const double result_abs, const double result_asc)
{
    err = m_fabs(err);
    if (result_asc != 0 && err != 0) {
        double scale = m_pow15_rescale(200 * err / result_asc);
        if (scale < 1)
            err = result_asc * scale;
        else
            err = result_asc;
    }
    if (result_abs > RIM_DBL_MIN / (50 * RIM_DBL_EPSILON)) {
        double min_err = 50 * RIM_DBL_EPSILON * result_abs;
        if (min_err > err)
            err = min_err;
    }
    return err;
}

#if !defined(RIMO_GK_PER_NODE)
/* The deterministic flavour forms QUADPACK's own terms -- w (f1 + f2) per symmetric pair of nodes, the centre alone --
 * and adds the 16 terms in the order of the wavefront kernel: pair k of the rule (abscissa xgk[k], k = 0..14; slot 15
 * is the centre) sits on lanes 2k and 2k + 1 of a half-wave, the first butterfly step (xor 1) is the pair sum
 * f1 + f2, and the remaining four steps (xor 2, 4, 8, 16) add the 16 weighted terms as a balanced binary tree over
 * the slots: adjacent slots, then pairs of slots, ...  (wave_qag.h wave_gk31; DESIGN.md "GK31 reduction order").
 * Only the ORDER of these 16 additions differs from qk.c's sequential loops; every term has the reference's bits. */
static double tree16(double *v)
{
    for (int width = 1; width < 16; width <<= 1)
        for (int i = 0; i < 16; i += 2 * width)
            v[i] = v[i] + v[i + width];
    return v[0];
}
#endif

void rimo_qk31(rimo_fn f, void *ctx, double a, double b,
               double *result, double *abserr, double *resabs, double *resasc)
{
    const double center = 0.5 * (a + b);
    const double half_length = 0.5 * (b - a);
    const double abs_half_length = m_fabs(half_length);
    double fv[32];
    double result_kronrod, result_gauss, result_abs, result_asc, mean, err;

    /* abscissae in ascending order; node j sits in lane j of the device rule */
    for (int j = 0; j < 31; j++)
        fv[j] = f(center + half_length * GK_X[j], ctx);
    fv[31] = 0.0;

#if defined(RIMO_GK_PER_NODE)
    if (0) {
#else
    /* QUADPACK order: centre, 7 Gauss pairs, 8 Kronrod-only pairs (xgk[k] = -GK_X[k]) */
    if (RIMO_LIT(RIMO_ATTR_GKSEQ)) {
#endif
        const double f_center = fv[15];
        result_gauss = f_center * GK_WG[15];
        result_kronrod = f_center * GK_WK[15];
        result_abs = m_fabs(result_kronrod);
        /* -DRIMO_GK_REVERSED (noise-floor control, tools/make_literal_fixtures.py): the same terms added in the
         * opposite order -- an equally legitimate evaluation of the same rule */
#ifdef RIMO_GK_REVERSED
#define GK_LOOP(j, n) for (int j = (n) - 1; j >= 0; j--)
#else
#define GK_LOOP(j, n) for (int j = 0; j < (n); j++)
#endif
        GK_LOOP(j, 7) {
            const int jtw = 2 * j + 1;
            const double fsum = fv[jtw] + fv[30 - jtw];
            result_gauss += GK_WG[jtw] * fsum;
            result_kronrod += GK_WK[jtw] * fsum;
            result_abs += GK_WK[jtw] * (m_fabs(fv[jtw]) + m_fabs(fv[30 - jtw]));
        }
        GK_LOOP(j, 8) {
            const int jtwm1 = 2 * j;
            result_kronrod += GK_WK[jtwm1] * (fv[jtwm1] + fv[30 - jtwm1]);
            result_abs += GK_WK[jtwm1] * (m_fabs(fv[jtwm1]) + m_fabs(fv[30 - jtwm1]));
        }
        mean = result_kronrod * 0.5;
        result_asc = GK_WK[15] * m_fabs(f_center - mean);
        GK_LOOP(j, 7) {
            const int jtw = 2 * j + 1;
            result_asc += GK_WK[jtw] * (m_fabs(fv[jtw] - mean) + m_fabs(fv[30 - jtw] - mean));
        }
        GK_LOOP(j, 8) {
            const int jtwm1 = 2 * j;
            result_asc += GK_WK[jtwm1] * (m_fabs(fv[jtwm1] - mean) + m_fabs(fv[30 - jtwm1] - mean));
        }
    }
#if defined(RIMO_GK_PER_NODE)
    else {
        /* The deterministic flavour as it was until round 2 (tools/nan_rootcause.py only): one product w f per NODE,
         * 32 terms in the butterfly order.  With subnormal samples every product rounds to whole quanta of 2^-1074, so
         * the products of small samples vanish where qk.c's pair terms w (f1 + f2) survive: the cause of the one-sided
         * NaNs of round 2. */
        double t[32];
#define TREE32(v) do { for (int w_ = 1; w_ < 32; w_ <<= 1) for (int i_ = 0; i_ < 32; i_ += 2 * w_) v[i_] = v[i_] + v[i_ + w_]; } while (0)
        for (int j = 0; j < 32; j++) t[j] = GK_WK[j] * fv[j];
        TREE32(t); result_kronrod = t[0];
        for (int j = 0; j < 32; j++) t[j] = GK_WG[j] * fv[j];
        TREE32(t); result_gauss = t[0];
        for (int j = 0; j < 32; j++) t[j] = GK_WK[j] * m_fabs(fv[j]);
        TREE32(t); result_abs = t[0];
        mean = result_kronrod * 0.5;
        for (int j = 0; j < 32; j++) t[j] = GK_WK[j] * m_fabs(fv[j] - mean);
        TREE32(t); result_asc = t[0];
    }
#else
    else {
        /* slot k < 15: the pair (node k, node 30 - k) of the ascending table, weight GK_WK[k] = GK_WK[30 - k]; slot 15: centre */
        double t[16];
        for (int k = 0; k < 15; k++) t[k] = GK_WK[k] * (fv[k] + fv[30 - k]);
        t[15] = GK_WK[15] * (fv[15] + 0.0);
        result_kronrod = tree16(t);
        for (int k = 0; k < 15; k++) t[k] = GK_WG[k] * (fv[k] + fv[30 - k]);
        t[15] = GK_WG[15] * (fv[15] + 0.0);
        result_gauss = tree16(t);
        for (int k = 0; k < 15; k++) t[k] = GK_WK[k] * (m_fabs(fv[k]) + m_fabs(fv[30 - k]));
        t[15] = GK_WK[15] * (m_fabs(fv[15]) + 0.0);
        result_abs = tree16(t);
        mean = result_kronrod * 0.5;
        for (int k = 0; k < 15; k++) t[k] = GK_WK[k] * (m_fabs(fv[k] - mean) + m_fabs(fv[30 - k] - mean));
        t[15] = GK_WK[15] * (m_fabs(fv[15] - mean) + 0.0);     /* the padding lane holds 0, not |0 - mean| */
        result_asc = tree16(t);
    }
#endif

    err = (result_kronrod - result_gauss) * half_length;
    result_kronrod *= half_length;
    result_abs *= abs_half_length;
    result_asc *= abs_half_length;

    *result = result_kronrod;
    *resabs = result_abs;
    *resasc = result_asc;
    *abserr = rescale_error(err, result_abs, result_asc);
}

/* ---- workspace (integration/workspace.c, util.c, qpsrt.c) -------------- */

rimo_workspace *rimo_workspace_alloc(size_t n)
{
    rimo_workspace *w = (rimo_workspace *) malloc(sizeof(rimo_workspace));
    w->alist = (double *) malloc(n * sizeof(double));
    w->blist = (double *) malloc(n * sizeof(double));
    w->rlist = (double *) malloc(n * sizeof(double));
    w->elist = (double *) malloc(n * sizeof(double));
    w->order = (size_t *) malloc(n * sizeof(size_t));
    w->level = (size_t *) malloc(n * sizeof(size_t));
    w->size = 0;
    w->limit = n;
    w->maximum_level = 0;
    w->nrmax = 0;
    w->i = 0;
    return w;
}

void rimo_workspace_free(rimo_workspace *w)
{
    if (!w) return;
    free(w->level); free(w->order); free(w->elist);
    free(w->rlist); free(w->blist); free(w->alist);
    free(w);
}

static void ws_initialise(rimo_workspace *w, double a, double b)
{
    w->size = 0;
    w->nrmax = 0;
    w->i = 0;
    w->alist[0] = a;
    w->blist[0] = b;
    w->rlist[0] = 0.0;
    w->elist[0] = 0.0;
    w->order[0] = 0;
    w->level[0] = 0;
    w->maximum_level = 0;
}

static void ws_set_initial_result(rimo_workspace *w, double result, double error)
{
    w->size = 1;
    w->rlist[0] = result;
    w->elist[0] = error;
}

/* qpsrt.c: maintain the descending ordering of error estimates */
static void ws_qpsrt(rimo_workspace *w)
{
    const size_t last = w->size - 1;
    const size_t limit = w->limit;
    double *elist = w->elist;
    size_t *order = w->order;
    double errmax, errmin;
    int i, k, top;
    size_t i_nrmax = w->nrmax;
    size_t i_maxerr = order[i_nrmax];

    if (last < 2) {
        order[0] = 0;
        order[1] = 1;
        w->i = i_maxerr;
        return;
    }

    errmax = elist[i_maxerr];

    while (i_nrmax > 0 && errmax > elist[order[i_nrmax - 1]]) {
        order[i_nrmax] = order[i_nrmax - 1];
        i_nrmax--;
    }

    if (last < (limit / 2 + 2))
        top = (int) last;
    else
        top = (int) (limit - last + 1);

    i = (int) i_nrmax + 1;
    while (i < top && errmax < elist[order[i]]) {
        order[i - 1] = order[i];
        i++;
    }
    order[i - 1] = i_maxerr;

    errmin = elist[last];
    k = top - 1;
    while (k > i - 2 && errmin >= elist[order[k]]) {
        order[k + 1] = order[k];
        k--;
    }
    order[k + 1] = last;

    i_maxerr = order[i_nrmax];
    w->i = i_maxerr;
    w->nrmax = i_nrmax;
}

static void ws_update(rimo_workspace *w, double a1, double b1, double area1, double error1,
                      double a2, double b2, double area2, double error2)
{
    const size_t i_max = w->i;
    const size_t i_new = w->size;
    const size_t new_level = w->level[i_max] + 1;

    if (error2 > error1) {
        w->alist[i_max] = a2;   /* blist[i_max] is already b2 */
        w->rlist[i_max] = area2;
        w->elist[i_max] = error2;
        w->level[i_max] = new_level;
        w->alist[i_new] = a1;
        w->blist[i_new] = b1;
        w->rlist[i_new] = area1;
        w->elist[i_new] = error1;
        w->level[i_new] = new_level;
    } else {
        w->blist[i_max] = b1;   /* alist[i_max] is already a1 */
        w->rlist[i_max] = area1;
        w->elist[i_max] = error1;
        w->level[i_max] = new_level;
        w->alist[i_new] = a2;
        w->blist[i_new] = b2;
        w->rlist[i_new] = area2;
        w->elist[i_new] = error2;
        w->level[i_new] = new_level;
    }
    w->size++;
    if (new_level > w->maximum_level)
        w->maximum_level = new_level;
    ws_qpsrt(w);
}

static int subinterval_too_small(double a1, double a2, double b2)
{
    const double e = RIM_DBL_EPSILON;
    const double u = RIM_DBL_MIN;
    double tmp = (1 + 100 * e) * (m_fabs(a2) + 1000 * u);
    return m_fabs(a1) <= tmp && m_fabs(b2) <= tmp;
}

/* investigation knob (rimo.h): per-bisection trace of the calling thread's next rimo_qag call */
static __thread double *t_trace = 0;
static __thread size_t t_trace_cap = 0, t_trace_rows = 0;
void rimo_set_qag_trace(double *buf, size_t cap_rows) { t_trace = buf; t_trace_cap = cap_rows; t_trace_rows = 0; }
size_t rimo_qag_trace_rows(void) { return t_trace_rows; }

/* investigation knob (environment RIMO_PICK_STATS=<min size>, read once; passive otherwise): for every quadrature that
 * ends with at least that many subintervals, how many ROUNDS it would have taken had the children of the M intervals with
 * the largest errors been evaluated together whenever the interval qag.c picks next is not among those already evaluated
 * (M = 2, 4, 8, 16) -- the measurement VERDICT round 3 asks for before an outer-level stash is built (item 6b): qag.c's
 * sequence of picks, sums and decisions is untouched, only the evaluation of children would be batched. */
#define PICK_M 4
static int pick_stats_min(void)
{
    static int v = -1;
    if (v < 0) { const char *e = getenv("RIMO_PICK_STATS"); v = e ? atoi(e) : 0; }
    return v;
}
typedef struct { double a[16], b[16]; int n; uint64_t rounds; } pick_set;
static void pick_step(pick_set *ps, int M, const rimo_workspace *w, double a_i, double b_i)
{
    for (int k = 0; k < ps->n; k++)
        if (ps->a[k] == a_i && ps->b[k] == b_i) {       /* its children are on file: no new round */
            ps->a[k] = ps->a[ps->n - 1]; ps->b[k] = ps->b[ps->n - 1]; ps->n--;
            return;
        }
    /* a new round: the M intervals with the largest errors (the picked one is the largest) */
    ps->rounds++;
    ps->n = 0;
    char *taken = (char *) calloc(w->size, 1);
    for (int m = 0; m < M && (size_t) m < w->size; m++) {
        size_t best = w->size;
        for (size_t k = 0; k < w->size; k++)
            if (!taken[k] && (best == w->size || w->elist[k] > w->elist[best])) best = k;
        if (best == w->size) break;
        taken[best] = 1;
        if (w->alist[best] == a_i && w->blist[best] == b_i) continue;      /* consumed right away */
        ps->a[ps->n] = w->alist[best]; ps->b[ps->n] = w->blist[best]; ps->n++;
    }
    free(taken);
}

/* qag.c */
int rimo_qag(rimo_fn f, void *ctx, double a, double b, double epsabs, double epsrel,
             size_t limit, rimo_workspace *w, double *result, double *abserr, uint64_t *gk_evals)
{
    double area, errsum;
    double result0, abserr0, resabs0, resasc0;
    double tolerance;
    size_t iteration = 0;
    int roundoff_type1 = 0, roundoff_type2 = 0, error_type = 0;
    double round_off;
    uint64_t nev = 0;

    ws_initialise(w, a, b);
    *result = 0;
    *abserr = 0;

    if (limit > w->limit)
        return 4; /* GSL_EINVAL */
    if (epsabs <= 0 && (epsrel < 50 * RIM_DBL_EPSILON || epsrel < 0.5e-28))
        return RIMO_EBADTOL;

    double *const trace = t_trace;          /* armed for this call only (nested calls run untraced) */
    t_trace = 0;
    rimo_qk31(f, ctx, a, b, &result0, &abserr0, &resabs0, &resasc0);
    nev++;
    ws_set_initial_result(w, result0, abserr0);
    if (trace && t_trace_cap) {
        double *r = trace;
        r[0] = a; r[1] = b; r[2] = result0; r[3] = abserr0; r[4] = resabs0; r[5] = resasc0;
        for (int k = 6; k < 14; k++) r[k] = 0.;
        t_trace_rows = 1;
    }

    tolerance = rim_max(epsabs, epsrel * m_fabs(result0));
    round_off = 50 * RIM_DBL_EPSILON * resabs0;

    if (abserr0 <= round_off && abserr0 > tolerance) {
        *result = result0;
        *abserr = abserr0;
        if (gk_evals) *gk_evals += nev;
        return RIMO_EROUND;
    } else if ((abserr0 <= tolerance && abserr0 != resasc0) || abserr0 == 0.0) {
        *result = result0;
        *abserr = abserr0;
        if (gk_evals) *gk_evals += nev;
        return RIMO_SUCCESS;
    } else if (limit == 1) {
        *result = result0;
        *abserr = abserr0;
        if (gk_evals) *gk_evals += nev;
        return RIMO_EMAXITER;
    }

    area = result0;
    errsum = abserr0;
    iteration = 1;
    const int pick_min = pick_stats_min();
    pick_set ps[PICK_M];
    uint64_t child_picks = 0;
    double prev_a = 0, prev_b = 0, prev_mid = 0;
    if (pick_min) for (int k = 0; k < PICK_M; k++) { ps[k].n = 0; ps[k].rounds = 0; }

    do {
        double a1, b1, a2, b2;
        double a_i, b_i, r_i, e_i;
        double area1 = 0, area2 = 0, area12 = 0;
        double error1 = 0, error2 = 0, error12 = 0;
        double resasc1, resasc2;
        double resabs1, resabs2;

        /* bisect the subinterval with the largest error estimate */
        a_i = w->alist[w->i];
        b_i = w->blist[w->i];
        r_i = w->rlist[w->i];
        e_i = w->elist[w->i];

        a1 = a_i;
        b1 = 0.5 * (a_i + b_i);
        a2 = b1;
        b2 = b_i;
        if (pick_min) {
            for (int k = 0; k < PICK_M; k++) pick_step(&ps[k], 2 << k, w, a_i, b_i);
            if (iteration > 1 && ((a_i == prev_a && b_i == prev_mid) || (a_i == prev_mid && b_i == prev_b))) child_picks++;
            prev_a = a_i; prev_b = b_i; prev_mid = b1;
        }

        rimo_qk31(f, ctx, a1, b1, &area1, &error1, &resabs1, &resasc1);
        rimo_qk31(f, ctx, a2, b2, &area2, &error2, &resabs2, &resasc2);
        nev += 2;

        area12 = area1 + area2;
        error12 = error1 + error2;

        errsum += (error12 - e_i);
        area += area12 - r_i;

        if (resasc1 != error1 && resasc2 != error2) {
            double delta = r_i - area12;
            if (m_fabs(delta) <= 1.0e-5 * m_fabs(area12) && error12 >= 0.99 * e_i)
                roundoff_type1++;
            if (iteration >= 10 && error12 > e_i)
                roundoff_type2++;
        }

        tolerance = rim_max(epsabs, epsrel * m_fabs(area));

        if (errsum > tolerance) {
            if (roundoff_type1 >= 6 || roundoff_type2 >= 20)
                error_type = 2; /* round off error */
            if (subinterval_too_small(a1, a2, b2))
                error_type = 3;
        }

        if (trace && t_trace_rows < t_trace_cap) {
            double *r = trace + 14 * t_trace_rows++;
            r[0] = a_i; r[1] = b_i; r[2] = r_i; r[3] = e_i; r[4] = area1; r[5] = error1; r[6] = resasc1;
            r[7] = area2; r[8] = error2; r[9] = resasc2; r[10] = roundoff_type1; r[11] = roundoff_type2;
            r[12] = errsum; r[13] = tolerance;
        }

        ws_update(w, a1, b1, area1, error1, a2, b2, area2, error2);

        iteration++;
    } while (iteration < limit && !error_type && errsum > tolerance);

    {
        double sum = 0;
        for (size_t k = 0; k < w->size; k++)
            sum += w->rlist[k];
        *result = sum;
    }
    *abserr = errsum;
    if (gk_evals) *gk_evals += nev;
    if (pick_min && w->size >= (size_t) pick_min)
        fprintf(stderr, "RIMO_PICK_STATS size %zu steps %zu limit %zu: next pick is a child of the interval just bisected in %.1f %% of the steps; "
                "rounds with the children of the top 2 / 4 / 8 / 16 intervals evaluated together: %llu %llu %llu %llu (steps per round %.2f %.2f %.2f %.2f)\n",
                w->size, iteration - 1, limit, 100. * (double) child_picks / (double) (iteration - 1),
                (unsigned long long) ps[0].rounds, (unsigned long long) ps[1].rounds, (unsigned long long) ps[2].rounds, (unsigned long long) ps[3].rounds,
                (double) (iteration - 1) / ps[0].rounds, (double) (iteration - 1) / ps[1].rounds, (double) (iteration - 1) / ps[2].rounds, (double) (iteration - 1) / ps[3].rounds);

    if (errsum <= tolerance)
        return RIMO_SUCCESS;
    else if (error_type == 2)
        return RIMO_EROUND;
    else if (error_type == 3)
        return RIMO_ESING;
    else if (iteration == limit)
        return RIMO_EMAXITER;
    else
        return RIMO_EFAILED;
}

int rimo_qag_gk31(rimo_fn f, void *ctx, double a, double b, double epsabs, double epsrel,
                  size_t limit, double *result, double *abserr, size_t *size_out, uint64_t *gk_evals)
{
    rimo_workspace *w = rimo_workspace_alloc(limit);
    int st = rimo_qag(f, ctx, a, b, epsabs, epsrel, limit, w, result, abserr, gk_evals);
    if (size_out) *size_out = w->size;
    rimo_workspace_free(w);
    return st;
}

/* ---- deriv/deriv.c ----------------------------------------------------- */

static void central_deriv(rimo_fn f, void *ctx, double x, double h,
                          double *result, double *abserr_round, double *abserr_trunc)
{
    double fm1 = f(x - h, ctx);
    double fp1 = f(x + h, ctx);
    double fmh = f(x - h / 2, ctx);
    double fph = f(x + h / 2, ctx);

    double r3 = 0.5 * (fp1 - fm1);
    double r5 = (4.0 / 3.0) * (fph - fmh) - (1.0 / 3.0) * r3;

    double e3 = (m_fabs(fp1) + m_fabs(fm1)) * RIM_DBL_EPSILON;
    double e5 = 2.0 * (m_fabs(fph) + m_fabs(fmh)) * RIM_DBL_EPSILON + e3;

    double dy = rim_max(m_fabs(r3 / h), m_fabs(r5 / h)) * (m_fabs(x) / h) * RIM_DBL_EPSILON;

    *result = r5 / h;
    *abserr_trunc = m_fabs((r5 - r3) / h);
    *abserr_round = m_fabs(e5 / h) + dy;
}

int rimo_deriv_central(rimo_fn f, void *ctx, double x, double h, double *result, double *abserr)
{
    double r_0, round, trunc, error;
    central_deriv(f, ctx, x, h, &r_0, &round, &trunc);
    error = round + trunc;

    if (round < trunc && (round > 0 && trunc > 0)) {
        double r_opt, round_opt, trunc_opt, error_opt;
        double h_opt = h * m_pow(round / (2.0 * trunc), 1.0 / 3.0);
        central_deriv(f, ctx, x, h_opt, &r_opt, &round_opt, &trunc_opt);
        error_opt = round_opt + trunc_opt;
        if (error_opt < error && m_fabs(r_opt - r_0) < 4.0 * error) {
            r_0 = r_opt;
            error = error_opt;
        }
    }

    *result = r_0;
    *abserr = error;
    return RIMO_SUCCESS;
}

/* gsl_sf_hyperg_2F1(a,b,c,x=1) for c-a-b > 0: Gauss's summation theorem,
 * exp(lnG(c) + lnG(c-a-b) - lnG(c-a) - lnG(c-b)).  Only ever called as
 * 2F1(1/2, -k/2; 3/2; 1) (pitchy_pl.rs:98, pitchy_kappa.rs:93), where all four
 * Gamma arguments are positive. */
double rimo_hyperg_2F1_at_1(double a, double b, double c)
{
    const double lc = m_lgamma(c);
    const double lcab = m_lgamma(c - a - b);
    const double lca = m_lgamma(c - a);
    const double lcb = m_lgamma(c - b);
    return m_exp(lc + lcab - lca - lcb);
}

#ifdef RIMO_ATTR
unsigned rimo_attr_mask = 0;
void rimo_set_attr_mask(unsigned mask) { rimo_attr_mask = mask; }
#else
void rimo_set_attr_mask(unsigned mask) { (void) mask; }
#endif

const char *rimo_build_flavour(void)
{
#if defined(RIMO_LIBM) && defined(RIMO_GK_REVERSED)
    return "libm+gsl-order-reversed";
#elif defined(RIMO_LIBM) && defined(__FP_FAST_FMA) && defined(RIMO_CONTRACT)
    return "libm+gsl-order+contracted";
#elif defined(RIMO_LIBM)
    return "libm+gsl-order";
#elif defined(RIMO_ATTR)
    return "attribution build (detmath+tree-order with run-time literal switches)";
#elif defined(RIMO_GK_PER_NODE)
    return "detmath+tree-order, per-node products (round 2)";
#else
    return "detmath+tree-order";
#endif
}
