/* oracle/iou_box3d_oracle.c -- TEST INFRASTRUCTURE (CPU oracle), never linked into the product.
 *
 * Plain-C restatement of the IoU3D path the reference evaluates with:
 *   box3d_overlap / _check_coplanar / _check_nonzero
 *        /root/reference/cubercnn/evaluation/omni3d_evaluation.py:65-86, 89-104, 106-166
 *   which calls pytorch3d._C.iou_box3d (omni3d_evaluation.py:155).
 *
 * pytorch3d is an UN-VENDORED dependency (README.md:60-65 of the reference names a conda
 * `pytorch3d` with no version; `_C.iou_box3d` exists from v0.6.0).  Its sources are not under
 * /root/reference and the package is not installed, so the algorithm below restates the
 * published CPU implementation (pytorch3d/csrc/iou_box3d/iou_utils.h + iou_box3d_cpu.cpp,
 * v0.7.x lineage): clip the 12 triangles of each box against the 6 face planes of the other,
 * drop box2 triangles coplanar with a box1 triangle, sum tetrahedra about the polyhedron
 * centre.  Constants kEpsilon=1e-8, dEpsilon=1e-3, aEpsilon=1e-4.
 *
 * PARITY PIN: there are no golden vectors for this path in the reference.  This
 * oracle is pinned by tests/test_iou3d_oracle.py against (i) analytic known answers and (ii) an
 * independent float64 half-space-intersection oracle (scipy) on generic pairs.  Parity with the
 * pytorch3d *binary* is unpinned.
 *
 * Arithmetic is float32 throughout, like vec3<float> upstream.  Build with -ffp-contract=off.
 */
#include <math.h>
#include <stdlib.h>
#include <string.h>

#define K_EPS 1e-8f
#define D_EPS 1e-3f
#define A_EPS 1e-4f

typedef struct { float x, y, z; } v3;
typedef struct { v3 v[3]; } tri_t;
typedef struct { v3 v[4]; } face_t;

static const int BOX_TRIS[12][3] = {
    {0, 1, 2}, {0, 3, 2}, {4, 5, 6}, {4, 6, 7}, {1, 5, 6}, {1, 6, 2},
    {0, 4, 7}, {0, 7, 3}, {3, 2, 6}, {3, 6, 7}, {0, 1, 5}, {0, 4, 5}};
static const int BOX_PLANES[6][4] = {
    {0, 1, 2, 3}, {3, 2, 6, 7}, {0, 1, 5, 4}, {0, 3, 7, 4}, {1, 2, 6, 5}, {4, 5, 6, 7}};

static inline v3 vsub(v3 a, v3 b) { v3 r = {a.x - b.x, a.y - b.y, a.z - b.z}; return r; }
static inline v3 vadd(v3 a, v3 b) { v3 r = {a.x + b.x, a.y + b.y, a.z + b.z}; return r; }
static inline v3 vscale(v3 a, float s) { v3 r = {a.x * s, a.y * s, a.z * s}; return r; }
static inline v3 vdiv(v3 a, float s) { v3 r = {a.x / s, a.y / s, a.z / s}; return r; }
static inline float vdot(v3 a, v3 b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
static inline v3 vcross(v3 a, v3 b) {
    v3 r = {a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x};
    return r;
}
static inline float vnorm(v3 a) { return sqrtf(vdot(a, a)); }

/* iou_utils.h: GetNormal -- unit normal of span(e0, e1) */
static v3 get_normal(v3 e0, v3 e1) {
    v3 n = vcross(e0, e1);
    return vdiv(n, fmaxf(vnorm(n), K_EPS));
}
/* iou_utils.h: TriNormal -- best pair of centre->vertex edges (max |cross|) */
static v3 tri_normal(const tri_t* t) {
    v3 ctr = vdiv(vadd(vadd(t->v[0], t->v[1]), t->v[2]), 3.0f);
    float best = -1.0f;
    v3 n = {0.f, 0.f, 0.f};
    for (int i = 0; i < 2; ++i)
        for (int j = i + 1; j < 3; ++j) {
            v3 a = vsub(t->v[i], ctr), b = vsub(t->v[j], ctr);
            float d = vnorm(vcross(a, b));
            if (d > best) { best = d; n = get_normal(a, b); }
        }
    return n;
}
static float tri_area(const tri_t* t) {
    v3 n = vcross(vsub(t->v[1], t->v[0]), vsub(t->v[2], t->v[0]));
    return vnorm(n) / 2.0f;
}
static v3 plane_center(const face_t* p) {
    v3 c = vadd(vadd(vadd(p->v[0], p->v[1]), p->v[2]), p->v[3]);
    return vdiv(c, 4.0f);
}
/* iou_utils.h: PlaneNormalDirection -- unit normal of a quad face pointing towards `center` */
static v3 plane_normal_direction(const face_t* p, v3 center) {
    v3 pc = plane_center(p);
    float best = -1.0f;
    v3 n = {0.f, 0.f, 0.f};
    for (int i = 0; i < 3; ++i)
        for (int j = i + 1; j < 4; ++j) {
            v3 a = vsub(p->v[i], pc), b = vsub(p->v[j], pc);
            float d = vnorm(vcross(a, b));
            if (d > best) { best = d; n = get_normal(a, b); }
        }
    float c = vdot(vsub(center, pc), n);
    if (c < 0.0f) n = vscale(n, -1.0f);
    return n;
}
/* iou_utils.h: BoxCenter / PolyhedronCenter / BoxVolume */
static v3 box_center(const float* b) {
    v3 c = {0.f, 0.f, 0.f};
    for (int t = 0; t < 8; ++t) { c.x += b[3 * t]; c.y += b[3 * t + 1]; c.z += b[3 * t + 2]; }
    return vdiv(c, 8.0f);
}
static v3 polyhedron_center(const tri_t* tris, int n) {
    v3 c = {0.f, 0.f, 0.f};
    for (int t = 0; t < n; ++t) {
        c.x += (tris[t].v[0].x + tris[t].v[1].x + tris[t].v[2].x) / 3.0f;
        c.y += (tris[t].v[0].y + tris[t].v[1].y + tris[t].v[2].y) / 3.0f;
        c.z += (tris[t].v[0].z + tris[t].v[1].z + tris[t].v[2].z) / 3.0f;
    }
    return vdiv(c, (float)n);
}
static float tris_volume(const tri_t* tris, int n, v3 ctr) {
    float vol = 0.f;
    for (int t = 0; t < n; ++t) {
        v3 a = vsub(tris[t].v[0], ctr), b = vsub(tris[t].v[1], ctr), c = vsub(tris[t].v[2], ctr);
        vol += fabsf(vdot(a, vcross(b, c))) / 6.0f;
    }
    return vol;
}
/* iou_utils.h: ArgMaxVerts -- farthest (tri vertex, other vertex) pair; `other` has n verts */
static void argmax_verts(const tri_t* t, const v3* other, int n, v3* a, v3* b) {
    float best = -1.0f;
    v3 z = {0.f, 0.f, 0.f};
    *a = z; *b = z;
    for (int i = 0; i < 3; ++i)
        for (int j = 0; j < n; ++j) {
            float d = vnorm(vsub(t->v[i], other[j]));
            if (d > best) { best = d; *a = t->v[i]; *b = other[j]; }
        }
}
static int is_coplanar_tri_plane(const tri_t* t, const face_t* p, v3 normal) {
    v3 nt = tri_normal(t);
    int check1 = fabsf(vdot(nt, normal)) > 1.0f - D_EPS;
    v3 a, b;
    argmax_verts(t, p->v, 4, &a, &b);
    v3 d = vsub(a, b);
    d = vdiv(d, fmaxf(vnorm(d), K_EPS));
    int check2 = (fabsf(vdot(d, normal)) < D_EPS) || (fabsf(vdot(nt, d)) < D_EPS);
    return check1 && check2;
}
static int is_coplanar_tri_tri(const tri_t* t1, const tri_t* t2) {
    v3 n1 = tri_normal(t1), n2 = tri_normal(t2);
    int check1 = fabsf(vdot(n1, n2)) > 1.0f - D_EPS;
    v3 a, b;
    argmax_verts(t1, t2->v, 3, &a, &b);
    v3 d = vsub(a, b);
    d = vdiv(d, fmaxf(vnorm(d), K_EPS));
    int check2 = (fabsf(vdot(d, n1)) < D_EPS) || (fabsf(vdot(d, n2)) < D_EPS);
    return check1 && check2;
}
static int is_inside(v3 pc, v3 normal, v3 pt) { return vdot(vsub(pt, pc), normal) >= 0.0f; }
static v3 plane_edge_intersection(v3 pc, v3 normal, v3 p0, v3 p1) {
    v3 direc = vsub(p1, p0);
    direc = vdiv(direc, fmaxf(vnorm(direc), K_EPS));
    v3 p = vdiv(vadd(p1, p0), 2.0f);
    if (fabsf(vdot(direc, normal)) >= D_EPS) {
        float top = -1.0f * vdot(vsub(p0, pc), normal);
        float bot = vdot(vsub(p1, p0), normal);
        float a = top / bot;
        p = vadd(p0, vscale(vsub(p1, p0), a));
    }
    return p;
}
/* ClipTriByPlane: writes 0..2 triangles to out, returns the count */
static int clip_tri_by_plane(const face_t* plane, v3 normal, const tri_t* t, tri_t* out) {
    v3 pc = plane_center(plane);
    v3 v0 = t->v[0], v1 = t->v[1], v2 = t->v[2];
    int in0 = is_inside(pc, normal, v0), in1 = is_inside(pc, normal, v1), in2 = is_inside(pc, normal, v2);
    if (is_coplanar_tri_plane(t, plane, normal)) { out[0] = *t; return 1; }
    if (in0 && in1 && in2) { out[0] = *t; return 1; }
    if (!in0 && !in1 && !in2) return 0;
    v3 vout, vi1, vi2, vin, vo1, vo2;
    int one_out = 0;
    if (in0 && in1 && !in2) { one_out = 1; vout = v2; vi1 = v0; vi2 = v1; }
    else if (in0 && !in1 && in2) { one_out = 1; vout = v1; vi1 = v0; vi2 = v2; }
    else if (!in0 && in1 && in2) { one_out = 1; vout = v0; vi1 = v1; vi2 = v2; }
    else if (in0 && !in1 && !in2) { vin = v0; vo1 = v1; vo2 = v2; }
    else if (!in0 && !in1 && in2) { vin = v2; vo1 = v0; vo2 = v1; }
    else { vin = v1; vo1 = v0; vo2 = v2; }
    if (one_out) {
        v3 p1 = plane_edge_intersection(pc, normal, vi1, vout);
        v3 p2 = plane_edge_intersection(pc, normal, vi2, vout);
        out[0].v[0] = vi1; out[0].v[1] = p1; out[0].v[2] = vi2;
        out[1].v[0] = vi2; out[1].v[1] = p1; out[1].v[2] = p2;
        return 2;
    }
    v3 p1 = plane_edge_intersection(pc, normal, vin, vo1);
    v3 p2 = plane_edge_intersection(pc, normal, vin, vo2);
    out[0].v[0] = vin; out[0].v[1] = p1; out[0].v[2] = p2;
    return 1;
}

#define ORACLE_MAX_TRIS 1024
static int g_max_tris_seen = 0;
int iou_box3d_oracle_max_tris(void) { return g_max_tris_seen; }
/* high-water mark of (box1 list + box2 list) after the same plane pass: what a kernel that keeps both clip directions in ONE
 * list (csrc/iou_box3d.hip) must hold.  Debugging aid like g_max_tris_seen, updated without synchronisation. */
static int g_max_joint_seen = 0;
static _Thread_local int g_pass_count[6];
int iou_box3d_oracle_max_joint_tris(void) { return g_max_joint_seen; }

static void box_tris(const float* b, tri_t* out) {
    for (int t = 0; t < 12; ++t)
        for (int k = 0; k < 3; ++k) {
            const float* p = b + 3 * BOX_TRIS[t][k];
            out[t].v[k].x = p[0]; out[t].v[k].y = p[1]; out[t].v[k].z = p[2];
        }
}
static void box_planes(const float* b, face_t* out) {
    for (int t = 0; t < 6; ++t)
        for (int k = 0; k < 4; ++k) {
            const float* p = b + 3 * BOX_PLANES[t][k];
            out[t].v[k].x = p[0]; out[t].v[k].y = p[1]; out[t].v[k].z = p[2];
        }
}
/* BoxIntersections: clip `tris` (12) successively by the six planes; returns count */
static int box_intersections(const tri_t* tris, const face_t* planes, v3 center, tri_t* out) {
    static _Thread_local tri_t bufA[ORACLE_MAX_TRIS], bufB[ORACLE_MAX_TRIS];      /* per thread: the OpenMP baseline calls this concurrently */
    tri_t* cur = bufA; tri_t* nxt = bufB;
    int n = 12;
    memcpy(cur, tris, 12 * sizeof(tri_t));
    for (int p = 0; p < 6; ++p) {
        v3 nrm = plane_normal_direction(&planes[p], center);
        int m = 0;
        for (int t = 0; t < n; ++t) {
            tri_t o[2];
            int k = clip_tri_by_plane(&planes[p], nrm, &cur[t], o);
            for (int q = 0; q < k && m < ORACLE_MAX_TRIS; ++q) nxt[m++] = o[q];
        }
        tri_t* tmp = cur; cur = nxt; nxt = tmp;
        n = m;
        if (n > g_max_tris_seen) g_max_tris_seen = n;
        g_pass_count[p] = n;
    }
    memcpy(out, cur, n * sizeof(tri_t));
    return n;
}

/* pytorch3d _C.iou_box3d (iou_box3d_cpu.cpp): boxes1 (N,8,3), boxes2 (M,8,3) -> vol, iou (N,M) */
void iou_box3d_oracle(const float* boxes1, int N, const float* boxes2, int M, float* vol_out, float* iou_out) {
    static _Thread_local tri_t i1[2 * ORACLE_MAX_TRIS], i2[ORACLE_MAX_TRIS];
    for (int a = 0; a < N; ++a) {
        const float* b1 = boxes1 + 24 * a;
        tri_t t1[12]; face_t p1[6];
        box_tris(b1, t1); box_planes(b1, p1);
        v3 c1 = box_center(b1);
        float vol1 = tris_volume(t1, 12, c1);
        for (int b = 0; b < M; ++b) {
            const float* b2 = boxes2 + 24 * b;
            tri_t t2[12]; face_t p2[6];
            box_tris(b2, t2); box_planes(b2, p2);
            v3 c2 = box_center(b2);
            float vol2 = tris_volume(t2, 12, c2);
            int n1 = box_intersections(t1, p2, c2, i1);
            int pass1[6];
            memcpy(pass1, g_pass_count, sizeof(pass1));
            int n2 = box_intersections(t2, p1, c1, i2);
            for (int f = 0; f < 6; ++f)
                if (pass1[f] + g_pass_count[f] > g_max_joint_seen) g_max_joint_seen = pass1[f] + g_pass_count[f];
            int n1_orig = n1;
            if (n2 > 0) {
                for (int q = 0; q < n2; ++q) {
                    int keep = 1;
                    for (int r = 0; r < n1_orig; ++r) {
                        if (is_coplanar_tri_tri(&i1[r], &i2[q]) && tri_area(&i1[r]) > A_EPS) { keep = 0; }
                    }
                    if (keep) i1[n1++] = i2[q];
                }
            }
            float vol = 0.f, iou = 0.f;
            if (n1 > 0) {
                v3 pc = polyhedron_center(i1, n1);
                vol = tris_volume(i1, n1, pc);
                iou = vol / (vol1 + vol2 - vol);
            }
            vol_out[(size_t)a * M + b] = vol;
            iou_out[(size_t)a * M + b] = iou;
        }
    }
}

/* omni3d_evaluation.py:65-86  _check_coplanar(boxes, eps): one 18-term dot product per box */
static int check_coplanar(const float* b, float eps) {
    float acc = 0.f;
    for (int p = 0; p < 6; ++p) {
        v3 v[4];
        for (int k = 0; k < 4; ++k) {
            const float* q = b + 3 * BOX_PLANES[p][k];
            v[k].x = q[0]; v[k].y = q[1]; v[k].z = q[2];
        }
        v3 e0 = vsub(v[1], v[0]), e1 = vsub(v[2], v[0]);
        e0 = vdiv(e0, fmaxf(vnorm(e0), 1e-12f));   /* F.normalize eps */
        e1 = vdiv(e1, fmaxf(vnorm(e1), 1e-12f));
        v3 n = vcross(e0, e1);
        n = vdiv(n, fmaxf(vnorm(n), 1e-12f));
        acc += vdot(vsub(v[3], v[0]), n);
    }
    return fabsf(acc) < eps;
}
/* omni3d_evaluation.py:89-104  _check_nonzero(boxes, eps) */
static int check_nonzero(const float* b, float eps) {
    for (int t = 0; t < 12; ++t) {
        v3 v[3];
        for (int k = 0; k < 3; ++k) {
            const float* q = b + 3 * BOX_TRIS[t][k];
            v[k].x = q[0]; v[k].y = q[1]; v[k].z = q[2];
        }
        float area = vnorm(vcross(vsub(v[1], v[0]), vsub(v[2], v[0]))) / 2.0f;
        if (!(area > eps)) return 0;
    }
    return 1;
}
/* omni3d_evaluation.py:106-166  box3d_overlap: iou with invalid dt rows zeroed */
void box3d_overlap_oracle(const float* dt, int N, const float* gt, int M, float eps_coplanar, float eps_nonzero, float* iou_out) {
    float* vol = (float*)malloc(sizeof(float) * (size_t)(N > 0 ? N : 1) * (size_t)(M > 0 ? M : 1));
    iou_box3d_oracle(dt, N, gt, M, vol, iou_out);
    for (int a = 0; a < N; ++a) {
        int ok = check_coplanar(dt + 24 * a, eps_coplanar) && check_nonzero(dt + 24 * a, eps_nonzero);
        if (!ok) for (int b = 0; b < M; ++b) iou_out[(size_t)a * M + b] = 0.f;
    }
    free(vol);
}
/* paired form used by the bench: pair p = (dt[p], gt[p]) */
void iou_box3d_pairs_oracle(const float* dt, const float* gt, int P, float* iou_out) {
    for (int p = 0; p < P; ++p) {
        float vol, iou;
        iou_box3d_oracle(dt + 24 * p, 1, gt + 24 * p, 1, &vol, &iou);
        iou_out[p] = iou;
    }
}

/* all-cores form of the paired loop (SURVEY.md 8d CPU baseline): pairs are independent; `g_max_tris_seen` is a debugging
 * high-water mark updated without synchronisation, irrelevant for the result */
void iou_box3d_pairs_oracle_omp(const float* dt, const float* gt, int P, float* iou_out) {
#pragma omp parallel for schedule(dynamic, 64)
    for (int p = 0; p < P; ++p) {
        float vol, iou;
        iou_box3d_oracle(dt + 24 * p, 1, gt + 24 * p, 1, &vol, &iou);
        iou_out[p] = iou;
    }
}

