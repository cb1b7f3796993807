/*
 * lrc_oracle.c -- CPU restatement of the LiDAR ray-cast path.  TEST INFRASTRUCTURE ONLY.
 *
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may build, load or call this
 * file.  The product (liblidarcast + the Python package) never does: it has no CPU path.
 *
 * What it restates
 *   The reference casts rays with open3d.t.geometry.RaycastingScene (Embree), a third-party
 *   dependency that is not vendored (requirements.txt:2, "open3d>=0.17.0", un-pinned) and is not
 *   installed here.  Call sites: raycast_engine/raycast_engine_cpu.py:46-51 and
 *   raycast_engine/raycast_engine_gpu_simple.py:41-46.  This file restates the published contract
 *   of that call -- closest hit, two-sided, tnear = 0 (exclusive), tfar = +inf, t parametric along
 *   the un-normalised direction, +inf / 0xFFFFFFFF on a miss, primitive id = triangle row -- with
 *   Embree 3's Moeller-Trumbore single-ray intersector (C = v0-O, R = C x D, den = Ng.D,
 *   U = R.e2, V = R.e1, T = Ng.C, sign-corrected; U,V >= 0, U+V <= |den|, T > 0; t = T/|den|).
 *
 * PARITY UNPINNED at the Embree boundary: the reference ships no test, golden vector or fixture for
 * this path, and Open3D cannot be run here, so the float32
 * bits of t cannot be pinned to Embree's.  What IS pinned: everything numpy around the cast
 * (oracle/np_oracle.py against goldens captured from the reference's own lidar/ package and
 * post-processing code, tests/golden/), and the analytic known-answer scenes in tests/.
 *
 * The hit definition is a pure function of (ray, triangle): every float32
 * operation and every fused multiply-add below is part of the specification, and the closest hit is
 * the lexicographic minimum of (t, triangle row).  orc_cast_brute evaluates it over all triangles;
 * orc_cast_bvh uses its own median-split BVH (not the product's SAH tree) and must agree exactly.
 *
 * Also here: orc_witness_f64, an INDEPENDENT double-precision statement of the closest two-sided hit (textbook
 * Moeller-Trumbore, no box clause), used by the tests to bound |t_float32 - t_exact| and to count rays on which the
 * float32 definition and exact geometry disagree; and orc_cast_bvh_diag, which counts how often the box clause acts.
 *
 * Build: see oracle/Makefile  (gcc -O2 -ffp-contract=off -shared -fPIC -pthread).
 */
#include <math.h>
#include <pthread.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#define ORC_CLONES __attribute__((target_clones("fma", "default")))
#define ORC_INL static inline __attribute__((always_inline))

#define TINY_DIR 1e-30f
#define PAD_REL_LO 0.999755859375f    /* 1 - 2^-12 */
#define PAD_REL_HI 1.000244140625f    /* 1 + 2^-12 */
#define PAD_ABS 1.52587890625e-05f    /* 2^-16 */

typedef struct { float x, y, z; } v3;

ORC_INL float fma_(float a, float b, float c) { return __builtin_fmaf(a, b, c); }
ORC_INL float dot3(v3 a, v3 b) { return fma_(a.z, b.z, fma_(a.y, b.y, a.x * b.x)); }
ORC_INL v3 cross3(v3 a, v3 b) {
    v3 r;
    r.x = fma_(a.y, b.z, -(a.z * b.y));
    r.y = fma_(a.z, b.x, -(a.x * b.z));
    r.z = fma_(a.x, b.y, -(a.y * b.x));
    return r;
}
ORC_INL v3 sub3(v3 a, v3 b) { v3 r = {a.x - b.x, a.y - b.y, a.z - b.z}; return r; }
ORC_INL float xorsign(float x, uint32_t s) {
    uint32_t u; memcpy(&u, &x, 4); u ^= s; memcpy(&x, &u, 4); return x;
}
ORC_INL float min2(float a, float b) { return a < b ? a : b; }
ORC_INL float max2(float a, float b) { return a > b ? a : b; }

typedef struct { float ix, iy, iz, ox, oy, oz; } slab_t;

ORC_INL float safe_inv(float d) {
    float a = fabsf(d);
    float s = a < TINY_DIR ? copysignf(TINY_DIR, d) : d;
    return 1.0f / s;
}
ORC_INL slab_t make_slab(v3 o, v3 d) {
    slab_t r;
    r.ix = safe_inv(d.x); r.iy = safe_inv(d.y); r.iz = safe_inv(d.z);
    r.ox = o.x * r.ix; r.oy = o.y * r.iy; r.oz = o.z * r.iz;
    return r;
}
/* padded [tn, tf] of the ray against the box; monotone in lo/hi so nested boxes give nested intervals */
ORC_INL void slab_interval(const slab_t* s, const float* lo, const float* hi, float* tn, float* tf) {
    float t0x = fma_(lo[0], s->ix, -s->ox), t1x = fma_(hi[0], s->ix, -s->ox);
    float t0y = fma_(lo[1], s->iy, -s->oy), t1y = fma_(hi[1], s->iy, -s->oy);
    float t0z = fma_(lo[2], s->iz, -s->oz), t1z = fma_(hi[2], s->iz, -s->oz);
    float nx = min2(t0x, t1x), fx = max2(t0x, t1x);
    float ny = min2(t0y, t1y), fy = max2(t0y, t1y);
    float nz = min2(t0z, t1z), fz = max2(t0z, t1z);
    float n = max2(max2(nx, ny), max2(nz, 0.0f));
    float f = min2(min2(fx, fy), fz);
    *tn = fma_(n, PAD_REL_LO, -PAD_ABS);
    *tf = fma_(f, PAD_REL_HI, PAD_ABS);
}

/* 1 and *t_out when the ray hits triangle (v0,v1,v2) */
ORC_INL int tri_hit(v3 o, v3 d, const slab_t* s, v3 v0, v3 v1, v3 v2, float* t_out) {
    v3 e1 = sub3(v0, v1);
    v3 e2 = sub3(v2, v0);
    v3 ng = cross3(e2, e1);
    v3 c = sub3(v0, o);
    v3 r = cross3(c, d);
    float den = dot3(ng, d);
    float aden = fabsf(den);
    uint32_t sgn; memcpy(&sgn, &den, 4); sgn &= 0x80000000u;
    float u = xorsign(dot3(r, e2), sgn);
    float v = xorsign(dot3(r, e1), sgn);
    float tt = xorsign(dot3(ng, c), sgn);
    if (!((den != 0.0f) && (u >= 0.0f) && (v >= 0.0f) && (u + v <= aden) && (tt > 0.0f))) return 0;
    float t = tt / aden;
    float lo[3], hi[3], tn, tf;
    lo[0] = min2(min2(v0.x, v1.x), v2.x); hi[0] = max2(max2(v0.x, v1.x), v2.x);
    lo[1] = min2(min2(v0.y, v1.y), v2.y); hi[1] = max2(max2(v0.y, v1.y), v2.y);
    lo[2] = min2(min2(v0.z, v1.z), v2.z); hi[2] = max2(max2(v0.z, v1.z), v2.z);
    slab_interval(s, lo, hi, &tn, &tf);
    if (!((tn <= t) && (t <= tf) && (t < INFINITY))) return (t < INFINITY) ? -1 : 0;   /* -1: rejected by the pad clause alone */
    *t_out = t;
    return 1;
}

/* finite-ray contract (include/lidarcast.h): a ray with a NaN or infinite component is never cast -- a miss */
ORC_INL int finite_ray(const float* r) {
    for (int k = 0; k < 6; ++k) {
        uint32_t u; memcpy(&u, &r[k], 4);
        if ((u & 0x7F800000u) == 0x7F800000u) return 0;
    }
    return 1;
}

ORC_INL v3 vert(const float* verts, uint32_t i) {
    v3 r = {verts[3 * (size_t)i], verts[3 * (size_t)i + 1], verts[3 * (size_t)i + 2]};
    return r;
}

/* ------------------------------------------------------------------------------------------------
 * brute force: the definition itself
 * ---------------------------------------------------------------------------------------------- */
ORC_CLONES
void orc_cast_brute(const float* verts, const uint32_t* tris, uint64_t T,
                    const float* rays6, uint64_t N, float* t_out, uint32_t* prim_out) {
    for (uint64_t i = 0; i < N; ++i) {
        const float* r = rays6 + 6 * i;
        v3 o = {r[0], r[1], r[2]}, d = {r[3], r[4], r[5]};
        slab_t s = make_slab(o, d);
        float best = INFINITY;
        uint32_t bp = 0xFFFFFFFFu;
        for (uint64_t k = 0; k < (finite_ray(r) ? T : 0); ++k) {
            float t;
            if (tri_hit(o, d, &s, vert(verts, tris[3 * k]), vert(verts, tris[3 * k + 1]),
                        vert(verts, tris[3 * k + 2]), &t) == 1) {
                if (t < best || (t == best && (uint32_t)k < bp)) { best = t; bp = (uint32_t)k; }
            }
        }
        t_out[i] = best;
        prim_out[i] = bp;
    }
}

/* ------------------------------------------------------------------------------------------------
 * the oracle's own BVH: median split on the widest centroid axis, one box per node, leaves <= 2
 * ---------------------------------------------------------------------------------------------- */
typedef struct {
    float lo[3], hi[3];
    int32_t left;     /* inner: index of left child (right = left + 1 is NOT assumed; see right) */
    int32_t right;
    uint32_t first, count;   /* leaf: count > 0, triangles order[first .. first+count) */
} onode;

typedef struct orc_bvh {
    const float* verts;
    const uint32_t* tris;
    uint64_t T;
    onode* nodes;
    uint32_t num_nodes;
    uint32_t* order;
    float* cent;     /* T x 3 centroids (build scratch, kept for simplicity) */
} orc_bvh;

static void tri_box(const orc_bvh* b, uint32_t k, float* lo, float* hi) {
    for (int a = 0; a < 3; ++a) {
        float x0 = b->verts[3 * (size_t)b->tris[3 * (size_t)k] + a];
        float x1 = b->verts[3 * (size_t)b->tris[3 * (size_t)k + 1] + a];
        float x2 = b->verts[3 * (size_t)b->tris[3 * (size_t)k + 2] + a];
        lo[a] = min2(min2(x0, x1), x2);
        hi[a] = max2(max2(x0, x1), x2);
    }
}

/* quickselect (Lomuto, middle pivot) on order[lo..hi) under the strict total order
 * (centroid[axis], index): afterwards order[nth] is in its sorted place.  Any outcome would still
 * give a valid BVH; the selection only balances it. */
static int key_lt(const orc_bvh* b, uint32_t q, uint32_t p, int axis) {
    float qv = b->cent[3 * (size_t)q + axis], pv = b->cent[3 * (size_t)p + axis];
    return qv < pv || (qv == pv && q < p);
}
static void select_nth(orc_bvh* b, uint32_t lo, uint32_t hi, uint32_t nth, int axis) {
    while (hi - lo > 1) {
        uint32_t m = lo + (hi - lo) / 2, tmp;
        tmp = b->order[m]; b->order[m] = b->order[hi - 1]; b->order[hi - 1] = tmp;
        uint32_t pivot = b->order[hi - 1], store = lo;
        for (uint32_t i = lo; i + 1 < hi; ++i) {
            if (key_lt(b, b->order[i], pivot, axis)) {
                tmp = b->order[i]; b->order[i] = b->order[store]; b->order[store] = tmp;
                ++store;
            }
        }
        tmp = b->order[store]; b->order[store] = b->order[hi - 1]; b->order[hi - 1] = tmp;
        if (nth == store) return;
        if (nth < store) hi = store; else lo = store + 1;
    }
}

/* Node numbering: a subtree over `count` triangles owns the index range [base, base + 2*count - 1) -- itself at `base`,
 * its left subtree right behind it, the right one after the left one's range.  Ranges of different subtrees are
 * disjoint, so the top of the tree can be built by several threads without any shared counter, and the result does not
 * depend on how many threads ran. */
typedef struct { orc_bvh* b; uint32_t first, count, base; int depth; } btask_t;
static void obuild(orc_bvh* b, uint32_t first, uint32_t count, uint32_t base, int depth);
static void* btask_main(void* p) {
    btask_t* t = (btask_t*)p;
    obuild(t->b, t->first, t->count, t->base, t->depth);
    return NULL;
}

static void obuild(orc_bvh* b, uint32_t first, uint32_t count, uint32_t base, int depth) {
    onode* n = &b->nodes[base];
    float clo[3] = {INFINITY, INFINITY, INFINITY}, chi[3] = {-INFINITY, -INFINITY, -INFINITY};
    for (int a = 0; a < 3; ++a) { n->lo[a] = INFINITY; n->hi[a] = -INFINITY; }
    for (uint32_t i = first; i < first + count; ++i) {
        float lo[3], hi[3];
        uint32_t k = b->order[i];
        tri_box(b, k, lo, hi);
        for (int a = 0; a < 3; ++a) {
            n->lo[a] = min2(n->lo[a], lo[a]); n->hi[a] = max2(n->hi[a], hi[a]);
            float c = b->cent[3 * (size_t)k + a];
            clo[a] = min2(clo[a], c); chi[a] = max2(chi[a], c);
        }
    }
    if (count <= 2) { n->first = first; n->count = count; n->left = n->right = -1; return; }
    int axis = 0;
    float e0 = chi[0] - clo[0], e1 = chi[1] - clo[1], e2 = chi[2] - clo[2];
    if (e1 > e0 && e1 >= e2) axis = 1; else if (e2 > e0 && e2 > e1) axis = 2;
    uint32_t half = count / 2;
    select_nth(b, first, first + count, first + half, axis);
    n->count = 0; n->first = 0;
    const uint32_t lbase = base + 1, rbase = base + 1 + (2 * half - 1);
    n->left = (int32_t)lbase; n->right = (int32_t)rbase;
    /* the first four levels fork: 16 subtrees side by side (the per-pose rebuild of the reference-faithful baseline) */
    pthread_t th;
    btask_t task = {b, first, half, lbase, depth + 1};
    int forked = depth < 4 && count > 20000 && pthread_create(&th, NULL, btask_main, &task) == 0;
    if (!forked) obuild(b, first, half, lbase, depth + 1);
    obuild(b, first + half, count - half, rbase, depth + 1);
    if (forked) pthread_join(th, NULL);
}

orc_bvh* orc_bvh_build(const float* verts, const uint32_t* tris, uint64_t T) {
    orc_bvh* b = (orc_bvh*)calloc(1, sizeof(orc_bvh));
    if (!b) return NULL;
    b->verts = verts; b->tris = tris; b->T = T;
    if (T == 0) return b;
    b->nodes = (onode*)malloc(sizeof(onode) * (2 * T));
    b->order = (uint32_t*)malloc(sizeof(uint32_t) * T);
    b->cent = (float*)malloc(sizeof(float) * 3 * T);
    if (!b->nodes || !b->order || !b->cent) { free(b->nodes); free(b->order); free(b->cent); free(b); return NULL; }
    for (uint64_t k = 0; k < T; ++k) {
        float lo[3], hi[3];
        b->order[k] = (uint32_t)k;
        tri_box(b, (uint32_t)k, lo, hi);
        for (int a = 0; a < 3; ++a) b->cent[3 * k + a] = 0.5f * lo[a] + 0.5f * hi[a];
    }
    obuild(b, 0, (uint32_t)T, 0, 0);
    b->num_nodes = (uint32_t)(2 * T - 1);
    return b;
}

void orc_bvh_free(orc_bvh* b) {
    if (!b) return;
    free(b->nodes); free(b->order); free(b->cent); free(b);
}

ORC_CLONES
static void cast_range(const orc_bvh* b, const float* rays6, uint64_t begin, uint64_t end,
                       float* t_out, uint32_t* prim_out, uint32_t* pad_rej_out) {
    int32_t stack[128];
    for (uint64_t i = begin; i < end; ++i) {
        const float* r = rays6 + 6 * i;
        v3 o = {r[0], r[1], r[2]}, d = {r[3], r[4], r[5]};
        slab_t s = make_slab(o, d);
        float best = INFINITY;
        uint32_t bp = 0xFFFFFFFFu, pad_rej = 0;
        int sp = 0;
        if (b->T && finite_ray(r)) stack[sp++] = 0;
        while (sp) {
            const onode* n = &b->nodes[stack[--sp]];
            float tn, tf;
            slab_interval(&s, n->lo, n->hi, &tn, &tf);
            if (!(tn <= tf && tn <= best)) continue;
            if (n->count) {
                for (uint32_t j = n->first; j < n->first + n->count; ++j) {
                    uint32_t k = b->order[j];
                    float t;
                    int h = tri_hit(o, d, &s, vert(b->verts, b->tris[3 * (size_t)k]),
                                    vert(b->verts, b->tris[3 * (size_t)k + 1]),
                                    vert(b->verts, b->tris[3 * (size_t)k + 2]), &t);
                    if (h == 1) {
                        if (t < best || (t == best && k < bp)) { best = t; bp = k; }
                    } else if (h < 0) {
                        ++pad_rej;
                    }
                }
            } else {
                stack[sp++] = n->right;
                stack[sp++] = n->left;
            }
        }
        t_out[i] = best;
        prim_out[i] = bp;
        if (pad_rej_out) pad_rej_out[i] = pad_rej;
    }
}

typedef struct {
    const orc_bvh* b; const float* rays6; uint64_t begin, end; float* t; uint32_t* prim; uint32_t* pad_rej;
    double* t64; double* margin64;     /* witness jobs */
} job_t;

static void* job_main(void* p) {
    job_t* j = (job_t*)p;
    cast_range(j->b, j->rays6, j->begin, j->end, j->t, j->prim, j->pad_rej);
    return NULL;
}

/* closest hit of N rays with `threads` pthreads (threads <= 1: in the calling thread) */
static void run_jobs(void* (*fn)(void*), job_t proto, uint64_t N, int threads) {
    if (threads <= 1 || N < 1024) { proto.begin = 0; proto.end = N; fn(&proto); return; }
    if (threads > 256) threads = 256;
    pthread_t th[256];
    job_t jobs[256];
    uint64_t chunk = (N + threads - 1) / threads;
    int started = 0;
    for (int k = 0; k < threads; ++k) {
        uint64_t a = (uint64_t)k * chunk, e = a + chunk;
        if (a >= N) break;
        if (e > N) e = N;
        jobs[k] = proto;
        jobs[k].begin = a; jobs[k].end = e;
        if (pthread_create(&th[k], NULL, fn, &jobs[k]) != 0) { fn(&jobs[k]); th[k] = 0; }
        started = k + 1;
    }
    for (int k = 0; k < started; ++k) if (th[k]) pthread_join(th[k], NULL);
}

void orc_cast_bvh(const orc_bvh* b, const float* rays6, uint64_t N, float* t_out, uint32_t* prim_out,
                  int threads) {
    job_t j = {b, rays6, 0, 0, t_out, prim_out, NULL, NULL, NULL};
    run_jobs(job_main, j, N, threads);
}

/* the same cast, plus per ray the number of triangles it visited that pass every Moeller-Trumbore condition
 * (den != 0, U,V >= 0, U+V <= |den|, T > 0, t finite) and are rejected ONLY by the clause "t inside the padded slab
 * interval of the triangle's own box" -- the one clause of the hit definition Embree does not have.  The count
 * depends on the traversal (culled subtrees are not visited); a total of 0 means the clause never acted. */
void orc_cast_bvh_diag(const orc_bvh* b, const float* rays6, uint64_t N, float* t_out, uint32_t* prim_out,
                       uint32_t* pad_rej_out, int threads) {
    job_t j = {b, rays6, 0, 0, t_out, prim_out, pad_rej_out, NULL, NULL};
    run_jobs(job_main, j, N, threads);
}

/* ------------------------------------------------------------------------------------------------
 * float64 witness: an INDEPENDENT statement of "closest two-sided hit" -- the textbook Moeller-Trumbore test
 * (edges from v0, barycentrics by division, no sign trick, no box clause) evaluated in double precision on the
 * float32 vertices and the float32 ray.  It shares nothing with tri_hit above but the mesh and the BVH topology
 * (boxes are re-tested in double, padded by 1e-9).  Tests use it to bound |t_float32 - t_exact| and to count rays on
 * which the float32 definition and double-precision geometry disagree about hit/miss or about the triangle.
 * margin = min(u, v, 1-u-v) of the winning hit (barycentric distance from the nearest edge), -1 on a miss.
 * ---------------------------------------------------------------------------------------------- */
static int tri_hit_f64(const double o[3], const double d[3], const float* a, const float* b_, const float* c,
                       double* t_out, double* margin_out) {
    double e1[3], e2[3], p[3], s[3], q[3];
    for (int k = 0; k < 3; ++k) { e1[k] = (double)b_[k] - a[k]; e2[k] = (double)c[k] - a[k]; s[k] = o[k] - a[k]; }
    p[0] = d[1] * e2[2] - d[2] * e2[1]; p[1] = d[2] * e2[0] - d[0] * e2[2]; p[2] = d[0] * e2[1] - d[1] * e2[0];
    double det = e1[0] * p[0] + e1[1] * p[1] + e1[2] * p[2];
    if (det == 0.0) return 0;
    double inv = 1.0 / det;
    double u = (s[0] * p[0] + s[1] * p[1] + s[2] * p[2]) * inv;
    q[0] = s[1] * e1[2] - s[2] * e1[1]; q[1] = s[2] * e1[0] - s[0] * e1[2]; q[2] = s[0] * e1[1] - s[1] * e1[0];
    double v = (d[0] * q[0] + d[1] * q[1] + d[2] * q[2]) * inv;
    double t = (e2[0] * q[0] + e2[1] * q[1] + e2[2] * q[2]) * inv;
    if (!(u >= 0.0 && v >= 0.0 && u + v <= 1.0 && t > 0.0 && t < INFINITY)) return 0;
    double w = 1.0 - u - v, m = u < v ? u : v;
    *t_out = t;
    *margin_out = m < w ? m : w;
    return 1;
}

static int box_hit_f64(const double o[3], const double d[3], const float* lo, const float* hi, double best) {
    double tn = 0.0, tf = best;
    for (int k = 0; k < 3; ++k) {
        double l = (double)lo[k] - 1e-9, h = (double)hi[k] + 1e-9;
        if (d[k] == 0.0) { if (o[k] < l || o[k] > h) return 0; continue; }
        double t0 = (l - o[k]) / d[k], t1 = (h - o[k]) / d[k];
        if (t0 > t1) { double x = t0; t0 = t1; t1 = x; }
        if (t0 > tn) tn = t0;
        if (t1 < tf) tf = t1;
    }
    return tn <= tf * (1.0 + 1e-12) + 1e-12;
}

static void* witness_main(void* p) {
    job_t* j = (job_t*)p;
    const orc_bvh* b = j->b;
    int32_t stack[128];
    for (uint64_t i = j->begin; i < j->end; ++i) {
        const float* r = j->rays6 + 6 * i;
        double o[3] = {r[0], r[1], r[2]}, d[3] = {r[3], r[4], r[5]};
        double best = INFINITY, bm = -1.0;
        uint32_t bp = 0xFFFFFFFFu;
        int sp = 0;
        if (b->T && finite_ray(r)) stack[sp++] = 0;
        while (sp) {
            const onode* n = &b->nodes[stack[--sp]];
            if (!box_hit_f64(o, d, n->lo, n->hi, best)) continue;
            if (n->count) {
                for (uint32_t q = n->first; q < n->first + n->count; ++q) {
                    uint32_t k = b->order[q];
                    double t, m;
                    if (tri_hit_f64(o, d, b->verts + 3 * (size_t)b->tris[3 * (size_t)k],
                                    b->verts + 3 * (size_t)b->tris[3 * (size_t)k + 1],
                                    b->verts + 3 * (size_t)b->tris[3 * (size_t)k + 2], &t, &m)) {
                        if (t < best || (t == best && k < bp)) { best = t; bp = k; bm = m; }
                    }
                }
            } else {
                stack[sp++] = n->right;
                stack[sp++] = n->left;
            }
        }
        j->t64[i] = best;
        j->prim[i] = bp;
        if (j->margin64) j->margin64[i] = bm;
    }
    return NULL;
}

void orc_witness_f64(const orc_bvh* b, const float* rays6, uint64_t N, double* t_out, uint32_t* prim_out,
                     double* margin_out, int threads) {
    job_t j = {b, rays6, 0, 0, NULL, prim_out, NULL, t_out, margin_out};
    run_jobs(witness_main, j, N, threads);
}

/* the witness test of ONE named triangle per ray (prim[i], 0xFFFFFFFF = skip): t and margin of that triangle alone,
 * +inf / -1 when double precision says the ray misses it.  Used to ask "what does exact geometry say about the
 * triangle the float32 definition chose". */
void orc_witness_tri_f64(const float* verts, const uint32_t* tris, const float* rays6, const uint32_t* prim,
                         uint64_t N, double* t_out, double* margin_out) {
    for (uint64_t i = 0; i < N; ++i) {
        t_out[i] = INFINITY; margin_out[i] = -1.0;
        if (prim[i] == 0xFFFFFFFFu) continue;
        const float* r = rays6 + 6 * i;
        double o[3] = {r[0], r[1], r[2]}, d[3] = {r[3], r[4], r[5]}, t, m;
        size_t k = prim[i];
        if (tri_hit_f64(o, d, verts + 3 * (size_t)tris[3 * k], verts + 3 * (size_t)tris[3 * k + 1],
                        verts + 3 * (size_t)tris[3 * k + 2], &t, &m)) { t_out[i] = t; margin_out[i] = m; }
    }
}

/* unit geometric normal of triangle rows prim[i] (0 for 0xFFFFFFFF): Ng / sqrt(Ng.Ng) */
ORC_CLONES
void orc_normals(const float* verts, const uint32_t* tris, const uint32_t* prim, uint64_t N,
                 float* normal3) {
    for (uint64_t i = 0; i < N; ++i) {
        float* q = normal3 + 3 * i;
        if (prim[i] == 0xFFFFFFFFu) { q[0] = q[1] = q[2] = 0.0f; continue; }
        size_t k = prim[i];
        v3 v0 = vert(verts, tris[3 * k]), v1 = vert(verts, tris[3 * k + 1]), v2 = vert(verts, tris[3 * k + 2]);
        v3 ng = cross3(sub3(v2, v0), sub3(v0, v1));
        float len = sqrtf(fma_(ng.z, ng.z, fma_(ng.y, ng.y, ng.x * ng.x)));
        q[0] = ng.x / len; q[1] = ng.y / len; q[2] = ng.z / len;
    }
}

int orc_has_fma(void) { return __builtin_cpu_supports("fma") ? 1 : 0; }
