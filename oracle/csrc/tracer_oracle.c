/* tracer_oracle.c -- brute-force closest-hit ray/mesh oracle in float64 (Moeller-Trumbore over every triangle).
 *
 * TEST INFRASTRUCTURE ONLY (never linked or called by the product path; see oracle/tracer_oracle.py).  It is the C restatement
 * of oracle/tracer_oracle.py::trace_bruteforce -- same arithmetic, same predicates, same tie rule (first triangle with the
 * smallest t) -- fast enough for the Stage-II parity tests at benchmark-like sizes (10^5 secondary rays), plus the per-ray
 * ambiguity flag of trace_bruteforce_margins.  The contract restated: raytracing/raytracer.py:21-55 and
 * network/renderer.py:719-729 of the reference (closest hit with t > 0, geometric normal from the winding, depth >= 10 = miss);
 * the third-party `_raytracing` binary itself is absent from /root/reference, so parity w.r.t. it is unpinned.
 *
 * Build (by __graft_entry__.build()):  gcc -O2 -fopenmp -shared -fPIC -o oracle/_build/libtracer_oracle.so oracle/csrc/tracer_oracle.c -lm
 */
#include <math.h>
#include <stdint.h>

static void cross3(const double* a, const double* b, double* c) {
    c[0] = a[1] * b[2] - a[2] * b[1];
    c[1] = a[2] * b[0] - a[0] * b[2];
    c[2] = a[0] * b[1] - a[1] * b[0];
}
static double dot3(const double* a, const double* b) { return a[0] * b[0] + a[1] * b[1] + a[2] * b[2]; }

/* verts [nV,3] float32, tris [nT,3] int32, o/d [n,3] float64 -> pos [n,3], nrm [n,3], depth [n] (float64), tri [n] int64,
 * amb [n] uint8 (may be NULL).  Returns 0. */
int tracer_oracle_trace(const float* verts, int nV, const int32_t* tris, int nT, const double* o, const double* d, int64_t n,
                        double miss_depth, double eps_edge, double eps_t, double* pos, double* nrm, double* depth, int64_t* tri,
                        uint8_t* amb) {
    (void)nV;
#pragma omp parallel for schedule(dynamic, 64)
    for (int64_t r = 0; r < n; ++r) {
        const double* ro = o + 3 * r;
        const double* rd = d + 3 * r;
        double best = INFINITY;
        int64_t bi = -1;
        /* pass 1: closest hit */
        for (int k = 0; k < nT; ++k) {
            double v0[3], e1[3], e2[3], p[3], tv[3], q[3];
            for (int a = 0; a < 3; ++a) {
                v0[a] = (double)verts[3 * tris[3 * k] + a];
                e1[a] = (double)verts[3 * tris[3 * k + 1] + a] - v0[a];
                e2[a] = (double)verts[3 * tris[3 * k + 2] + a] - v0[a];
                tv[a] = ro[a] - v0[a];
            }
            cross3(rd, e2, p);
            const double det = dot3(e1, p);
            if (!(fabs(det) > 1e-20)) continue;
            const double inv = 1.0 / det;
            const double u = dot3(tv, p) * inv;
            cross3(tv, e1, q);
            const double v = dot3(q, rd) * inv;
            const double t = dot3(e2, q) * inv;
            if (u >= 0 && u <= 1 && v >= 0 && u + v <= 1 && t > 0 && t < miss_depth && t < best) { best = t; bi = k; }
        }
        const double dep = bi >= 0 ? best : miss_depth;
        depth[r] = dep;
        tri[r] = bi;
        for (int a = 0; a < 3; ++a) { pos[3 * r + a] = ro[a] + dep * rd[a]; nrm[3 * r + a] = 0.0; }
        if (bi >= 0) {
            double v0[3], e1[3], e2[3], nn[3];
            for (int a = 0; a < 3; ++a) {
                v0[a] = (double)verts[3 * tris[3 * bi] + a];
                e1[a] = (double)verts[3 * tris[3 * bi + 1] + a] - v0[a];
                e2[a] = (double)verts[3 * tris[3 * bi + 2] + a] - v0[a];
            }
            cross3(e1, e2, nn);
            const double l = sqrt(dot3(nn, nn));
            for (int a = 0; a < 3; ++a) nrm[3 * r + a] = nn[a] / l;
        }
        if (!amb) continue;
        /* pass 2: could a float32 tracer legitimately answer differently?  an edge within eps_edge (barycentric units) of a
         * candidate that is not farther than the closest hit, or a candidate within eps_t of the ray origin */
        uint8_t flag = 0;
        for (int k = 0; k < nT && !flag; ++k) {
            double v0[3], e1[3], e2[3], p[3], tv[3], q[3];
            for (int a = 0; a < 3; ++a) {
                v0[a] = (double)verts[3 * tris[3 * k] + a];
                e1[a] = (double)verts[3 * tris[3 * k + 1] + a] - v0[a];
                e2[a] = (double)verts[3 * tris[3 * k + 2] + a] - v0[a];
                tv[a] = ro[a] - v0[a];
            }
            cross3(rd, e2, p);
            const double det = dot3(e1, p);
            if (!(fabs(det) > 1e-20)) continue;
            const double inv = 1.0 / det;
            const double u = dot3(tv, p) * inv;
            cross3(tv, e1, q);
            const double v = dot3(q, rd) * inv;
            const double t = dot3(e2, q) * inv;
            const double outside = fmax(fmax(-u, -v), u + v - 1.0);          /* <= 0 inside */
            if (fabs(outside) < eps_edge && t > -eps_t && t < dep + eps_t) flag = 1;
            if (outside < eps_edge && fabs(t) < eps_t) flag = 1;
        }
        amb[r] = flag;
    }
    return 0;
}
