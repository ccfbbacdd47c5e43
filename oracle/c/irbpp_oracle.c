/*
 * ORACLE (test infrastructure, not product code) -- plain C restatement of the reference's
 * no-physics packing environment, one bin per handle.  It follows the same reference lines as
 * the numpy oracle (oracle/space.py, oracle/cvtools.py, oracle/contours.py, oracle/packing.py,
 * which cite them one by one) and is cross-checked against that oracle and against the golden
 * vectors generated from the reference's own Python (tests/test_c_oracle.py).  Only tests/,
 * __graft_entry__.smoke() and bench.py's cpu_baseline leg may load it.
 *
 *   Space.get_possible_position   environment/physics0/space.py:98-129
 *   heightmap update              space.py:213 (closed form of space.py:75-94)
 *   convexHulls / getConvexHullActions / find_convex_vetex      cvTools.py:40-102
 *   cv2.findContours(RETR_TREE, CHAIN_APPROX_SIMPLE), cv2.approxPolyDP(c,1,True)
 *                                 OpenCV 4.4 contours.cpp / approx.cpp (restated; parity unpinned)
 *   cur_observation / step / reset / get_action_candidates      binPhy.py:128-337
 *   LoadItemCreator               IRcreator.py:74-103
 *
 * Build: gcc -O2 -ffp-contract=off -fno-fast-math -shared -fPIC (see oracle/Makefile).
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#define MAXR 8
#define MAXA 32                              /* action cells a side: 16 at resolutionA 0.02, 32 at 0.01 (space.py:24) */
#define MAXK 16

typedef struct {
    int fx, fy;
    double ext[3];
    const double *T, *B, *mH, *mB;          /* [fx][fy] row-major */
} orc_table;

typedef struct {
    int R, S, K, Hx, Hy, Ax, Ay, step;
    double resA, resH, resZ, bin[3], scale_z, ibin_z, bin_vol;
    int n_shapes;
    orc_table *tab;                          /* [n_shapes][R] */
    double *volume;
    int *seq, n_traj, seq_len, stride;
    long traj_index;
    int item_index;
    double *hm;                              /* [Hx][Hy] */
    double posz[MAXR][MAXA][MAXA], poszv[MAXR][MAXA][MAXA], mask[MAXR][MAXA][MAXA];
    int queue[MAXK + 4], qlen;
    int next_item, order_action, choose_item;
    double *cand;                            /* [S][5] */
    int item_idx;
    int packed_ids[1024];
    double *pool;                            /* copies of all tables (NULL: borrowed from the caller, orc_create_borrowed) */
    int borrowed;                            /* tables and trajectories belong to the caller and outlive the handle */
} orc_env;

static double round6(double x) { return nearbyint(x * 1e6) / 1e6; }

/* numpy floor_divide on float64 (npy_divmod) */
static double np_floor_divide(double a, double b) {
    double mod = fmod(a, b), div = (a - mod) / b, fd;
    if (mod != 0.0) { if ((b < 0) != (mod < 0)) { mod += b; div -= 1.0; } }
    if (div != 0.0) { fd = floor(div); if (div - fd > 0.5) fd += 1.0; }
    else fd = copysign(0.0, a / b);
    return fd;
}

/* ---------------------------------------------------------------- item queue (IRcreator.py) */
static void creator_generate(orc_env *e) {
    long row = e->traj_index % e->n_traj;
    if (row < 0) row += e->n_traj;
    e->queue[e->qlen++] = e->item_index < e->seq_len ? e->seq[row * e->seq_len + e->item_index] : -1;
    e->item_index++;
}
static void creator_reset(orc_env *e) { e->qlen = 0; e->traj_index += e->stride; e->item_index = 0; }
static void creator_preview(orc_env *e, int n) { while (e->qlen < n) creator_generate(e); }
static void creator_pop(orc_env *e, int idx) {
    for (int i = idx; i < e->qlen - 1; ++i) e->queue[i] = e->queue[i + 1];
    e->qlen--;
}

/* ---------------------------------------------------------------- space.py:98-129 */
static void get_possible_position(orc_env *e, int item) {
    for (int r = 0; r < e->R; ++r)
        for (int X = 0; X < e->Ax; ++X)
            for (int Y = 0; Y < e->Ay; ++Y) { e->posz[r][X][Y] = 1e3; e->mask[r][X][Y] = 0.0; }
    if (item >= 0) {
        for (int r = 0; r < e->R; ++r) {
            const orc_table *t = &e->tab[item * e->R + r];
            const double bx = round6(t->ext[0]), by = round6(t->ext[1]), bz = round6(t->ext[2]);
            const int oh_x = (int)ceil(bx / e->resH), oh_y = (int)ceil(by / e->resH);
            const int oa_x = (int)ceil(bx / e->resA), oa_y = (int)ceil(by / e->resA);
            for (int X = 0; X < e->Ax - oa_x + 1; ++X)
                for (int Y = 0; Y < e->Ay - oa_y + 1; ++Y) {
                    const int cx = X * e->step, cy = Y * e->step;
                    double m = -INFINITY;
                    for (int i = 0; i < oh_x; ++i)
                        for (int j = 0; j < oh_y; ++j) {
                            const double v = (e->hm[(cx + i) * e->Hy + cy + j] - t->B[i * t->fy + j]) * t->mB[i * t->fy + j];
                            if (v > m) m = v;
                        }
                    if (round6(m + bz - e->bin[2]) <= 0) e->mask[r][X][Y] = 1.0;
                    e->posz[r][X][Y] = m;
                }
        }
    }
    for (int r = 0; r < e->R; ++r)
        for (int X = 0; X < e->Ax; ++X)
            for (int Y = 0; Y < e->Ay; ++Y)
                e->poszv[r][X][Y] = e->mask[r][X][Y] == 0.0 ? 1e3 : e->posz[r][X][Y];
}

/* ---------------------------------------------------------------- OpenCV restatement */
static const int DX[8] = {1, 1, 0, -1, -1, -1, 0, 1}, DY[8] = {0, -1, -1, -1, 0, 1, 1, 1};

typedef struct { int n; int x[4096], y[4096]; } contour_t;

/* icvFetchContourEx, CHAIN_APPROX_SIMPLE, on the padded label image */
static void fetch_contour(int img[MAXA + 2][MAXA + 2], int x0, int y0, int is_hole, int nbd, contour_t *c) {
    int s_end = is_hole ? 0 : 4, s = s_end, x1, y1;
    c->n = 0;
    do { s = (s - 1) & 7; x1 = x0 + DX[s]; y1 = y0 + DY[s]; } while (img[y1][x1] == 0 && s != s_end);
    if (s == s_end) { img[y0][x0] = -nbd; c->x[0] = x0; c->y[0] = y0; c->n = 1; return; }
    int x3 = x0, y3 = y0, x4 = x0, y4 = y0, prev_s = s ^ 4, px = x0, py = y0;
    for (;;) {
        s_end = s;
        while (s < 15) { ++s; x4 = x3 + DX[s & 7]; y4 = y3 + DY[s & 7]; if (img[y4][x4] != 0) break; }
        s &= 7;
        if ((unsigned)(s - 1) < (unsigned)s_end) img[y3][x3] = -nbd;
        else if (img[y3][x3] == 1) img[y3][x3] = nbd;
        if (s != prev_s) { if (c->n < 4096) { c->x[c->n] = px; c->y[c->n] = py; } c->n++; }
        prev_s = s;
        px += DX[s]; py += DY[s];
        if (x4 == x0 && y4 == y0 && x3 == x1 && y3 == y1) break;
        x3 = x4; y3 = y4; s = (s + 4) & 7;
    }
}

/* approxPolyDP_<int>(closed, eps = 1); returns the polygon in (ox, oy) */
static int approx_poly_dp(const contour_t *c, int *ox, int *oy) {
    const int count0 = c->n;
    static __thread int sx_[8192], se_[8192];
    int count = count0, new_count = 0, top = 0, pos = 0, right_start = 0, le_eps = 0, spx = 0, spy = 0;
    if (count == 0) return 0;
    for (int it = 0; it < 3; ++it) {
        double max_dist = 0;
        pos = (pos + right_start) % count;
        spx = c->x[pos]; spy = c->y[pos];
        if (++pos >= count) pos = 0;
        for (int j = 1; j < count; ++j) {
            const double dx = c->x[pos] - spx, dy = c->y[pos] - spy, dist = dx * dx + dy * dy;
            if (++pos >= count) pos = 0;
            if (dist > max_dist) { max_dist = dist; right_start = j; }
        }
        le_eps = max_dist <= 1.0;
    }
    if (!le_eps) {
        const int s0 = pos % count, far = (right_start + s0) % count;
        sx_[top] = far; se_[top++] = s0;
        sx_[top] = s0; se_[top++] = far;
    } else { ox[new_count] = spx; oy[new_count++] = spy; }
    while (top > 0) {
        const int s_start = sx_[--top], s_end = se_[top];
        int le, split = 0;
        pos = s_start;
        spx = c->x[pos]; spy = c->y[pos];
        if (++pos >= count) pos = 0;
        if (pos != s_end) {
            const double dx = c->x[s_end] - spx, dy = c->y[s_end] - spy;
            double max_dist = 0;
            while (pos != s_end) {
                const double dist = fabs((c->y[pos] - spy) * dx - (c->x[pos] - spx) * dy);
                if (dist > max_dist) { max_dist = dist; split = pos; }
                if (++pos >= count) pos = 0;
            }
            le = max_dist * max_dist <= 1.0 * (dx * dx + dy * dy);
        } else le = 1;
        if (le) { ox[new_count] = spx; oy[new_count++] = spy; }
        else { sx_[top] = split; se_[top++] = s_end; sx_[top] = s_start; se_[top++] = split; }
    }
    {   /* clean-up, in place */
        const int cnt = new_count;
        int p2 = cnt - 1, stx = ox[p2], sty = oy[p2], wpos, ptx, pty;
        if (++p2 >= cnt) p2 = 0;
        wpos = p2; ptx = ox[p2]; pty = oy[p2];
        if (++p2 >= cnt) p2 = 0;
        for (int i = 0; i < cnt && new_count > 2; ++i) {
            const int ex = ox[p2], ey = oy[p2];
            if (++p2 >= cnt) p2 = 0;
            const double dx = ex - stx, dy = ey - sty;
            const double dist = fabs((ptx - stx) * dy - (pty - sty) * dx);
            const double inner = (double)(ptx - stx) * (ex - ptx) + (double)(pty - sty) * (ey - pty);
            if (dist * dist <= 0.5 * 1.0 * (dx * dx + dy * dy) && dx != 0 && dy != 0 && inner >= 0) {
                new_count--;
                ox[wpos] = stx = ex; oy[wpos] = sty = ey;
                if (++wpos >= cnt) wpos = 0;
                ptx = ox[p2]; pty = oy[p2];
                if (++p2 >= cnt) p2 = 0;
                i++;
                continue;
            }
            ox[wpos] = stx = ptx; oy[wpos] = sty = pty;
            if (++wpos >= cnt) wpos = 0;
            ptx = ex; pty = ey;
        }
    }
    return new_count;
}

/* convexHulls (cvTools.py:77-102) for one rotation: marks the candidate cells in hit[row][col] */
static void convex_hulls(orc_env *e, int r, unsigned char hit[MAXA][MAXA]) {
    int mapInt[MAXA][MAXA], levels[MAXA * MAXA], nlev = 0;
    static __thread contour_t c;
    static __thread int ox[4096], oy[4096];
    for (int X = 0; X < e->Ax; ++X)
        for (int Y = 0; Y < e->Ay; ++Y) {
            mapInt[X][Y] = e->mask[r][X][Y] == 0.0 ? -1 : (int)np_floor_divide(e->poszv[r][X][Y], e->resZ);
            int seen = 0;
            for (int k = 0; k < nlev; ++k) if (levels[k] == mapInt[X][Y]) seen = 1;
            if (!seen) levels[nlev++] = mapInt[X][Y];
        }
    for (int li = 0; li < nlev; ++li) {
        const int h = levels[li];
        if (h == -1) continue;
        int img[MAXA + 2][MAXA + 2];
        memset(img, 0, sizeof(img[0]) * (size_t)(e->Ax + 2));      /* rows 0 .. Ax + 1 of the padded image */
        for (int X = 0; X < e->Ax; ++X)
            for (int Y = 0; Y < e->Ay; ++Y) img[X + 1][Y + 1] = mapInt[X][Y] == h;
        int nbd = 1;
        for (int y = 1; y <= e->Ax; ++y) {
            int prev = 0;
            for (int x = 1; x <= e->Ay; ++x) {
                int p = img[y][x];
                if (p == prev) continue;
                int is_hole = 0, start = 0;
                if (prev == 0 && p == 1) start = 1;
                else if (p == 0 && prev >= 1) { is_hole = 1; start = 1; }
                if (start) {
                    ++nbd;
                    fetch_contour(img, is_hole ? x - 1 : x, y, is_hole, nbd, &c);
                    if (!is_hole) {                               /* find_out_contour keeps outer borders */
                        const int m = approx_poly_dp(&c, ox, oy);
                        for (int i = 0; i < m; ++i) {             /* find_convex_vetex */
                            int keep = 1;
                            if (m > 3) {
                                const int ax = ox[(i + m - 1) % m], ay = oy[(i + m - 1) % m];
                                const int cx = ox[(i + 1) % m], cy = oy[(i + 1) % m];
                                keep = (ox[i] - ax) * (cy - ay) - (oy[i] - ay) * (cx - ax) < 0;
                            }
                            if (keep) hit[oy[i] - 1][ox[i] - 1] = 1;   /* row = cv y, col = cv x */
                        }
                    }
                    p = img[y][x];
                }
                prev = p;
            }
        }
    }
}

/* ---------------------------------------------------------------- binPhy.cur_observation */
typedef struct { double key; int idx; } sort_item;
static int cmp_stable(const void *a, const void *b) {
    const sort_item *x = a, *y = b;
    if (x->key < y->key) return -1;
    if (x->key > y->key) return 1;
    return x->idx - y->idx;
}

static void cur_observation(orc_env *e, int gen_item, double *obs) {
    const int Hc = e->Hx * e->Hy, S = e->S;
    if (!e->choose_item) {
        if (gen_item) { creator_preview(e, 1); e->next_item = e->queue[0]; }
        get_possible_position(e, e->next_item);
        static __thread double all[MAXR * MAXA * MAXA][5];
        int n = 0;
        for (int r = 0; r < e->R; ++r) {                          /* getConvexHullActions */
            unsigned char hit[MAXA][MAXA];
            memset(hit, 0, sizeof(hit[0]) * (size_t)e->Ax);
            convex_hulls(e, r, hit);
            for (int col = 0; col < e->Ay; ++col)                 /* np.unique: sorted by (x = col, y = row) */
                for (int row = 0; row < e->Ax; ++row)
                    if (hit[row][col]) {
                        all[n][0] = r; all[n][1] = row; all[n][2] = col;
                        all[n][3] = e->poszv[r][row][col]; all[n][4] = e->mask[r][row][col];
                        ++n;
                    }
        }
        memset(e->cand, 0, sizeof(double) * S * 5);
        if (n > 0 && n <= S) memcpy(e->cand, all, sizeof(double) * n * 5);
        else if (n > S) {
            static __thread sort_item it[MAXR * MAXA * MAXA];
            for (int i = 0; i < n; ++i) { it[i].key = all[i][3]; it[i].idx = i; }
            qsort(it, n, sizeof(sort_item), cmp_stable);
            for (int i = 0; i < S; ++i) memcpy(e->cand + i * 5, all[it[i].idx], sizeof(double) * 5);
        } else {                                                  /* binPhy.py:217-225 */
            static __thread sort_item it[MAXR * MAXA * MAXA];
            const int total = e->R * e->Ax * e->Ay;
            for (int i = 0; i < total; ++i) { it[i].key = (&e->poszv[0][0][0])[(i / (e->Ax * e->Ay)) * MAXA * MAXA + ((i / e->Ay) % e->Ax) * MAXA + i % e->Ay]; it[i].idx = i; }
            qsort(it, total, sizeof(sort_item), cmp_stable);
            for (int i = 0; i < S && i < total; ++i) {
                const int idx = it[i].idx, r = idx / (e->Ax * e->Ay), X = (idx / e->Ay) % e->Ax, Y = idx % e->Ay;
                double *row = e->cand + i * 5;
                row[0] = r; row[1] = X; row[2] = Y; row[3] = e->bin[2]; row[4] = e->mask[r][X][Y];
            }
        }
        memcpy(obs, e->cand, sizeof(double) * S * 5);
        memset(obs + 5 * S, 0, sizeof(double) * 9);
        obs[5 * S] = e->next_item;
        memcpy(obs + 5 * S + 9, e->hm, sizeof(double) * Hc);
    } else {
        creator_preview(e, e->K);
        for (int i = 0; i < e->K; ++i) obs[i] = e->queue[i];
        memcpy(obs + e->K, e->hm, sizeof(double) * Hc);
    }
}

/* ---------------------------------------------------------------- public API */
static orc_env *create_env(int R, int S, int K, double resA, double resH, double resZ, const double *bin, double scale_z,
                           int n_shapes, const double *extents, const double *volumes, const int *dims,
                           const int64_t *offsets, int64_t pool_len, const double *T, const double *B, const double *mH,
                           const double *mB, const int *seq, int n_traj, int seq_len, int first_traj, int stride, int borrowed) {
    orc_env *e = calloc(1, sizeof(orc_env));
    e->borrowed = borrowed;
    e->R = R; e->S = S; e->K = K; e->resA = resA; e->resH = resH; e->resZ = resZ; e->scale_z = scale_z;
    for (int i = 0; i < 3; ++i) e->bin[i] = bin[i];
    e->bin_vol = bin[0] * bin[1] * bin[2];
    e->ibin_z = round6(bin[2] * scale_z);
    e->step = (int)(resA / resH);
    e->Hx = (int)ceil(bin[0] / resH); e->Hy = (int)ceil(bin[1] / resH);
    e->Ax = (int)ceil(bin[0] / resA); e->Ay = (int)ceil(bin[1] / resA);
    e->n_shapes = n_shapes;
    if (!borrowed) {
        e->pool = malloc(sizeof(double) * 4 * pool_len);
        memcpy(e->pool, T, sizeof(double) * pool_len);
        memcpy(e->pool + pool_len, B, sizeof(double) * pool_len);
        memcpy(e->pool + 2 * pool_len, mH, sizeof(double) * pool_len);
        memcpy(e->pool + 3 * pool_len, mB, sizeof(double) * pool_len);
        T = e->pool; B = e->pool + pool_len; mH = e->pool + 2 * pool_len; mB = e->pool + 3 * pool_len;
    }
    e->tab = calloc((size_t)n_shapes * R, sizeof(orc_table));
    for (int i = 0; i < n_shapes * R; ++i) {
        orc_table *t = &e->tab[i];
        t->fx = dims[2 * i]; t->fy = dims[2 * i + 1];
        for (int k = 0; k < 3; ++k) t->ext[k] = extents[3 * i + k];
        t->T = T + offsets[i]; t->B = B + offsets[i];
        t->mH = mH + offsets[i]; t->mB = mB + offsets[i];
    }
    e->volume = malloc(sizeof(double) * n_shapes);
    memcpy(e->volume, volumes, sizeof(double) * n_shapes);
    if (borrowed) {
        e->seq = (int *)seq;
    } else {
        e->seq = malloc(sizeof(int) * (size_t)n_traj * seq_len);
        memcpy(e->seq, seq, sizeof(int) * (size_t)n_traj * seq_len);
    }
    e->n_traj = n_traj; e->seq_len = seq_len; e->stride = stride; e->traj_index = (long)first_traj - stride;
    e->hm = calloc((size_t)e->Hx * e->Hy, sizeof(double));
    e->cand = calloc((size_t)S * 5, sizeof(double));
    e->choose_item = K > 1;
    return e;
}

orc_env *orc_create(int R, int S, int K, double resA, double resH, double resZ, const double *bin, double scale_z,
                    int n_shapes, const double *extents, const double *volumes, const int *dims,
                    const int64_t *offsets, int64_t pool_len, const double *T, const double *B, const double *mH,
                    const double *mB, const int *seq, int n_traj, int seq_len, int first_traj, int stride) {
    return create_env(R, S, K, resA, resH, resZ, bin, scale_z, n_shapes, extents, volumes, dims, offsets, pool_len, T, B, mH, mB,
                      seq, n_traj, seq_len, first_traj, stride, 0);
}

/* The same environment on tables and trajectories that stay the CALLER's (read-only, alive until orc_destroy): thousands
 * of bins of one data set share one copy (the 64 x 64 data set's tables are 54 MB). */
orc_env *orc_create_borrowed(int R, int S, int K, double resA, double resH, double resZ, const double *bin, double scale_z,
                             int n_shapes, const double *extents, const double *volumes, const int *dims,
                             const int64_t *offsets, int64_t pool_len, const double *T, const double *B, const double *mH,
                             const double *mB, const int *seq, int n_traj, int seq_len, int first_traj, int stride) {
    return create_env(R, S, K, resA, resH, resZ, bin, scale_z, n_shapes, extents, volumes, dims, offsets, pool_len, T, B, mH, mB,
                      seq, n_traj, seq_len, first_traj, stride, 1);
}

void orc_destroy(orc_env *e) {
    if (!e) return;
    if (!e->borrowed) { free(e->pool); free(e->seq); }
    free(e->tab); free(e->volume); free(e->hm); free(e->cand); free(e);
}

int orc_obs_len(const orc_env *e, int which) {
    const int loc = 5 * e->S + 9 + e->Hx * e->Hy;
    return which == 0 ? (e->K > 1 ? e->K + e->Hx * e->Hy : loc) : loc;
}

void orc_reset(orc_env *e, double *obs) {                        /* binPhy.py:128-147 */
    memset(e->hm, 0, sizeof(double) * e->Hx * e->Hy);
    creator_reset(e);
    e->item_idx = 0;
    cur_observation(e, 1, obs);
}

void orc_get_action_candidates(orc_env *e, int order_action, double *obs) {   /* binPhy.py:161-169 */
    e->next_item = e->queue[order_action];
    e->choose_item = 0;
    cur_observation(e, 0, obs);
    e->choose_item = 1;
    e->order_action = order_action;
}

double orc_get_ratio(const orc_env *e) {                         /* binPhy.py:149-153 */
    double tot = 0;
    for (int i = 0; i < e->item_idx; ++i) tot += e->volume[e->packed_ids[i]];
    return tot / e->bin_vol;
}

/* binPhy.py:248-337, no-physics branch.  Returns done; obs is the observation step() returns
 * (the caller applies the worker's auto-reset, shmem_vec_env.py:142-144). */
int orc_step(orc_env *e, int action, double *obs, double *reward, int *counter, double *ratio) {
    const double *row = e->cand + (size_t)action * 5;
    const int rot = (int)row[0], lx = (int)row[1], ly = (int)row[2];
    const int item = e->next_item;
    int success = item >= 0;
    if (success) {                                               /* prejudge (:238-245) */
        const orc_table *t = &e->tab[item * e->R + rot];
        const double tx = round6(lx * e->resA), ty = round6(ly * e->resA);
        double msum = 0;
        for (int r = 0; r < e->R; ++r) for (int X = 0; X < e->Ax; ++X) for (int Y = 0; Y < e->Ay; ++Y) msum += e->mask[r][X][Y];
        if (round6(tx + t->ext[0] - e->bin[0]) > 0 || round6(ty + t->ext[1] - e->bin[1]) > 0 || msum == 0) success = 0;
    }
    const double height = e->posz[rot][lx][ly];
    if (success) {                                               /* Interface.simulateHeight */
        const orc_table *t = &e->tab[item * e->R + rot];
        const double maxc = height * e->scale_z + t->ext[2] * e->scale_z;
        if (round6(maxc - e->ibin_z) > 0) success = 0;
    }
    if (!success) {
        *reward = 0.0; *counter = e->item_idx; *ratio = orc_get_ratio(e);
        cur_observation(e, 1, obs);
        return 1;
    }
    {   /* place_item: np.maximum(window, (T + z) * maskH) */
        const orc_table *t = &e->tab[item * e->R + rot];
        const int X = lx * e->step, Y = ly * e->step;
        for (int i = 0; i < t->fx; ++i)
            for (int j = 0; j < t->fy; ++j) {
                const double v = (t->T[i * t->fy + j] + height) * t->mH[i * t->fy + j];
                double *h = &e->hm[(X + i) * e->Hy + Y + j];
                if (v > *h) *h = v;
            }
    }
    e->packed_ids[e->item_idx] = item;
    *reward = (e->volume[item] / e->bin_vol) * 10;
    e->item_idx++;
    creator_pop(e, e->order_action);
    creator_generate(e);
    *counter = -1; *ratio = -1.0;
    cur_observation(e, 1, obs);
    return 0;
}

/* debug access for cross-checks */
void orc_get_grids(const orc_env *e, double *posz, double *mask) {
    int k = 0;
    for (int r = 0; r < e->R; ++r) for (int X = 0; X < e->Ax; ++X) for (int Y = 0; Y < e->Ay; ++Y) { posz[k] = e->posz[r][X][Y]; mask[k++] = e->mask[r][X][Y]; }
}
void orc_get_heightmap(const orc_env *e, double *hm) { memcpy(hm, e->hm, sizeof(double) * e->Hx * e->Hy); }
void orc_set_heightmap(orc_env *e, const double *hm) { memcpy(e->hm, hm, sizeof(double) * e->Hx * e->Hy); }

/* ---------------------------------------------------------------- many bins per call (parity runs at thousands of bins)
 * The same calls for n independent handles, dealt to `threads` host threads (joined before returning: nothing outlives
 * the call), results written straight into the caller's [n][obs_len] block.  orc_step_many applies the worker's
 * auto-reset (shmem_vec_env.py:141-144): a finished bin's row holds the next episode's first observation. */
#include <pthread.h>
typedef struct {
    orc_env **envs; int lo, hi, what; const int *arg; double *obs; size_t stride; double *rew; int *done, *counter; double *ratio;
} orc_job;
static void *orc_job_run(void *p) {
    orc_job *j = p;
    for (int i = j->lo; i < j->hi; ++i) {
        double *o = j->obs + (size_t)i * j->stride;
        if (j->what == 0) orc_reset(j->envs[i], o);
        else if (j->what == 1) orc_get_action_candidates(j->envs[i], j->arg[i], o);
        else {
            j->done[i] = orc_step(j->envs[i], j->arg[i], o, &j->rew[i], &j->counter[i], &j->ratio[i]);
            if (j->done[i]) orc_reset(j->envs[i], o);
        }
    }
    return NULL;
}
static void orc_many(orc_env **envs, int n, int threads, int what, const int *arg, double *obs, size_t stride, double *rew,
                     int *done, int *counter, double *ratio) {
    if (threads < 1) threads = 1;
    if (threads > 64) threads = 64;
    if (threads > n) threads = n > 0 ? n : 1;
    orc_job jobs[64]; pthread_t tid[64];
    for (int t = 0; t < threads; ++t) {
        jobs[t] = (orc_job){envs, (int)((long)n * t / threads), (int)((long)n * (t + 1) / threads), what, arg, obs, stride, rew, done, counter, ratio};
        if (t > 0 && pthread_create(&tid[t], NULL, orc_job_run, &jobs[t]) != 0) { orc_job_run(&jobs[t]); tid[t] = 0; }
    }
    orc_job_run(&jobs[0]);
    for (int t = 1; t < threads; ++t) if (tid[t]) pthread_join(tid[t], NULL);
}
void orc_reset_many(orc_env **envs, int n, int threads, double *obs, long stride) { orc_many(envs, n, threads, 0, NULL, obs, (size_t)stride, NULL, NULL, NULL, NULL); }
void orc_get_action_candidates_many(orc_env **envs, int n, int threads, const int *order_actions, double *obs, long stride) {
    orc_many(envs, n, threads, 1, order_actions, obs, (size_t)stride, NULL, NULL, NULL, NULL);
}
void orc_step_many(orc_env **envs, int n, int threads, const int *actions, double *obs, long stride, double *rew, int *done,
                   int *counter, double *ratio) {
    orc_many(envs, n, threads, 2, actions, obs, (size_t)stride, rew, done, counter, ratio);
}
