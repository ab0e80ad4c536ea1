// trace_stats.cpp -- host model of the Stage-II BVH traversal (nero_amd/csrc/bvh.hip): how many node fetches / triangle tests a ray
// needs and how many LOOP ITERATIONS a 64-ray wavefront needs (the slowest lane decides), for the binary tree of trace_kernel and
// for a 4-wide tree collapsed from it.  No GPU: this sizes the tracer redesign before any kernel is written.
//   g++ -O2 -o trace_stats trace_stats.cpp && ./trace_stats v.bin nV f.bin nT o.bin d.bin nR
#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

struct Node2 { float lmin[3], lmax[3], rmin[3], rmax[3]; int left, right; };
struct Tri { float v0[3], e1[3], e2[3]; };
static const int NONE = -(1 << 30);

struct Builder {
    const float* V; const int* F;
    std::vector<int> order; std::vector<float> cen, bmin, bmax; std::vector<Node2> nodes;
    void tri_box(int lo, int hi, float* mn, float* mx) const {
        for (int a = 0; a < 3; ++a) { mn[a] = 1e30f; mx[a] = -1e30f; }
        for (int i = lo; i < hi; ++i) { const int t = order[i]; for (int a = 0; a < 3; ++a) { mn[a] = std::min(mn[a], bmin[t * 3 + a]); mx[a] = std::max(mx[a], bmax[t * 3 + a]); } }
    }
    int build(int lo, int hi, int leaf_max) {
        const int n = hi - lo;
        if (n <= leaf_max) return -(lo * 8 + n) - 1;
        float cmn[3] = {1e30f, 1e30f, 1e30f}, cmx[3] = {-1e30f, -1e30f, -1e30f};
        for (int i = lo; i < hi; ++i) for (int a = 0; a < 3; ++a) { cmn[a] = std::min(cmn[a], cen[order[i] * 3 + a]); cmx[a] = std::max(cmx[a], cen[order[i] * 3 + a]); }
        int axis = 0;
        if (cmx[1] - cmn[1] > cmx[axis] - cmn[axis]) axis = 1;
        if (cmx[2] - cmn[2] > cmx[axis] - cmn[axis]) axis = 2;
        const int mid = lo + n / 2;
        std::nth_element(order.begin() + lo, order.begin() + mid, order.begin() + hi, [&](int a, int b) { return cen[a * 3 + axis] < cen[b * 3 + axis]; });
        const int me = (int)nodes.size();
        nodes.emplace_back();
        Node2 nd;
        tri_box(lo, mid, nd.lmin, nd.lmax);
        tri_box(mid, hi, nd.rmin, nd.rmax);
        nd.left = build(lo, mid, leaf_max);
        nd.right = build(mid, hi, leaf_max);
        nodes[me] = nd;
        return me;
    }
};

static bool box_hit(const float* mn, const float* mx, const float* o, const float* inv, float tbest, float& tn) {
    float t0 = 0.f, t1 = tbest;
    for (int a = 0; a < 3; ++a) {
        float ta = (mn[a] - o[a]) * inv[a], tb = (mx[a] - o[a]) * inv[a];
        t0 = std::max(t0, std::min(ta, tb)); t1 = std::min(t1, std::max(ta, tb));
    }
    tn = t0;
    return t0 <= t1;
}
static int leaf_test(const std::vector<Tri>& tris, int ref, const float* o, const float* d, float& tbest, int& best) {
    const int code = -ref - 1, start = code >> 3, count = code & 7;
    for (int i = 0; i < count; ++i) {
        const Tri& t = tris[start + i];
        const float px = d[1] * t.e2[2] - d[2] * t.e2[1], py = d[2] * t.e2[0] - d[0] * t.e2[2], pz = d[0] * t.e2[1] - d[1] * t.e2[0];
        const float det = t.e1[0] * px + t.e1[1] * py + t.e1[2] * pz;
        if (std::fabs(det) < 1e-20f) continue;
        const float inv = 1.0f / det;
        const float tx = o[0] - t.v0[0], ty = o[1] - t.v0[1], tz = o[2] - t.v0[2];
        const float u = (tx * px + ty * py + tz * pz) * inv;
        if (u < 0.f || u > 1.f) continue;
        const float qx = ty * t.e1[2] - tz * t.e1[1], qy = tz * t.e1[0] - tx * t.e1[2], qz = tx * t.e1[1] - ty * t.e1[0];
        const float v = (d[0] * qx + d[1] * qy + d[2] * qz) * inv;
        if (v < 0.f || u + v > 1.f) continue;
        const float tt = (t.e2[0] * qx + t.e2[1] * qy + t.e2[2] * qz) * inv;
        if (tt > 0.f && tt < tbest) { tbest = tt; best = start + i; }
    }
    return count;
}

// 4-wide node collapsed from the binary tree: up to 4 children (boxes + refs)
struct Node4 { float mn[4][3], mx[4][3]; int child[4]; int n; };

struct Stat { long long rays = 0, nodes = 0, tris = 0, hits = 0, wave_iters = 0, wave_tri_slots = 0, waves = 0, stack_max = 0, pushes = 0; };

template <class F> static void read_bin(const char* p, std::vector<F>& v, size_t n) {
    v.resize(n);
    FILE* f = fopen(p, "rb");
    if (!f || fread(v.data(), sizeof(F), n, f) != n) { fprintf(stderr, "cannot read %s\n", p); exit(1); }
    fclose(f);
}

int main(int argc, char** argv) {
    if (argc < 8) return 1;
    const int nV = atoi(argv[2]), nT = atoi(argv[4]), nR = atoi(argv[7]);
    std::vector<float> V, O, D; std::vector<int> F;
    read_bin(argv[1], V, (size_t)nV * 3); read_bin(argv[3], F, (size_t)nT * 3); read_bin(argv[5], O, (size_t)nR * 3); read_bin(argv[6], D, (size_t)nR * 3);
    for (int leaf_max : {4, 2}) {
        Builder B; B.V = V.data(); B.F = F.data();
        B.order.resize(nT); B.cen.resize((size_t)nT * 3); B.bmin.resize((size_t)nT * 3); B.bmax.resize((size_t)nT * 3);
        for (int t = 0; t < nT; ++t) {
            B.order[t] = t;
            for (int a = 0; a < 3; ++a) {
                float mn = 1e30f, mx = -1e30f, c = 0.f;
                for (int k = 0; k < 3; ++k) { const float x = V[(size_t)F[t * 3 + k] * 3 + a]; mn = std::min(mn, x); mx = std::max(mx, x); c += x; }
                B.bmin[t * 3 + a] = mn; B.bmax[t * 3 + a] = mx; B.cen[t * 3 + a] = c / 3.f;
            }
        }
        const int root = B.build(0, nT, leaf_max);
        std::vector<Tri> T(nT);
        for (int i = 0; i < nT; ++i) {
            const int t = B.order[i];
            const float *a = &V[(size_t)F[t * 3] * 3], *b = &V[(size_t)F[t * 3 + 1] * 3], *c = &V[(size_t)F[t * 3 + 2] * 3];
            for (int k = 0; k < 3; ++k) { T[i].v0[k] = a[k]; T[i].e1[k] = b[k] - a[k]; T[i].e2[k] = c[k] - a[k]; }
        }
        // ---- collapse into 4-wide nodes: a child that is an inner node is replaced by ITS two children
        std::vector<Node4> N4;
        std::vector<int> map4(B.nodes.size(), -1);
        struct Rec { static int go(const std::vector<Node2>& n2, std::vector<Node4>& n4, int cur) {
            const int me = (int)n4.size(); n4.emplace_back();
            Node4 q; q.n = 0;
            auto add = [&](const float* mn, const float* mx, int ref) { memcpy(q.mn[q.n], mn, 12); memcpy(q.mx[q.n], mx, 12); q.child[q.n] = ref; ++q.n; };
            const Node2& a = n2[cur];
            const float* bmn[2] = {a.lmin, a.rmin}; const float* bmx[2] = {a.lmax, a.rmax}; const int ch[2] = {a.left, a.right};
            for (int s = 0; s < 2; ++s) {
                if (ch[s] >= 0) { const Node2& g = n2[ch[s]]; add(g.lmin, g.lmax, g.left); add(g.rmin, g.rmax, g.right); }
                else add(bmn[s], bmx[s], ch[s]);
            }
            for (int k = 0; k < q.n; ++k) if (q.child[k] >= 0) q.child[k] = go(n2, n4, q.child[k]);
            n4[me] = q;
            return me;
        } };
        const int root4 = Rec::go(B.nodes, N4, root);
        printf("leaf<=%d: binary tree %zu nodes (%.1f MB at 64 B), 4-wide %zu nodes (%.1f MB at 128 B), triangles %.1f MB at 48 B / %.1f MB at 36 B\n", leaf_max,
               B.nodes.size(), B.nodes.size() * 64 / 1e6, N4.size(), N4.size() * 128 / 1e6, nT * 48 / 1e6, nT * 36 / 1e6);

        // ---- binary traversal as trace_kernel, wave by wave
        Stat s2, s4;
        std::vector<float> depth2(nR), depth4(nR);
        for (int w0 = 0; w0 < nR; w0 += 64) {
            long long it_max = 0;
            std::vector<std::vector<int>> tri_per_iter(64);
            for (int l = 0; l < 64 && w0 + l < nR; ++l) {
                const int r = w0 + l;
                const float* o = &O[(size_t)r * 3]; const float* d = &D[(size_t)r * 3];
                float inv[3];
                for (int a = 0; a < 3; ++a) inv[a] = 1.0f / (std::fabs(d[a]) > 1e-20f ? d[a] : (d[a] < 0.f ? -1e-20f : 1e-20f));
                float tbest = 10.f; int best = -1; int stack[64]; int sp = 0; int cur = root; long long it = 0;
                while (cur != NONE) {
                    const Node2& nd = B.nodes[cur];
                    ++it; ++s2.nodes;
                    int tcount = 0;
                    float tl, tr;
                    const bool hl = box_hit(nd.lmin, nd.lmax, o, inv, tbest, tl), hr = box_hit(nd.rmin, nd.rmax, o, inv, tbest, tr);
                    int next = NONE, first = nd.left, second = nd.right; bool hf = hl, hs = hr;
                    if (hl && hr && tr < tl) { first = nd.right; second = nd.left; }
                    if (!hl) { first = nd.right; hf = hr; hs = false; }
                    if (hf) { if (first < 0) tcount += leaf_test(T, first, o, d, tbest, best); else next = first; }
                    if (hs) { if (second < 0) tcount += leaf_test(T, second, o, d, tbest, best); else if (next == NONE) next = second; else { stack[sp++] = second; ++s2.pushes; s2.stack_max = std::max<long long>(s2.stack_max, sp); } }
                    if (next == NONE && sp > 0) next = stack[--sp];
                    cur = next;
                    s2.tris += tcount;
                    tri_per_iter[l].push_back(tcount);
                }
                it_max = std::max(it_max, it);
                ++s2.rays; s2.hits += best >= 0; depth2[r] = tbest;
            }
            s2.wave_iters += it_max; ++s2.waves;
            for (long long k = 0; k < it_max; ++k) { int m = 0; for (int l = 0; l < 64; ++l) if (k < (long long)tri_per_iter[l].size()) m = std::max(m, tri_per_iter[l][k]); s2.wave_tri_slots += m; }
        }
        // ---- 4-wide traversal: test the (up to) 4 boxes, visit hits nearest first, leaves tested at once
        for (int w0 = 0; w0 < nR; w0 += 64) {
            long long it_max = 0;
            std::vector<std::vector<int>> tri_per_iter(64);
            for (int l = 0; l < 64 && w0 + l < nR; ++l) {
                const int r = w0 + l;
                const float* o = &O[(size_t)r * 3]; const float* d = &D[(size_t)r * 3];
                float inv[3];
                for (int a = 0; a < 3; ++a) inv[a] = 1.0f / (std::fabs(d[a]) > 1e-20f ? d[a] : (d[a] < 0.f ? -1e-20f : 1e-20f));
                float tbest = 10.f; int best = -1; int stack[96]; float stack_t[96]; int sp = 0; int cur = root4; long long it = 0;
                if (root < 0) cur = NONE;
                while (cur != NONE) {
                    const Node4& nd = N4[cur];
                    ++it; ++s4.nodes;
                    int tcount = 0;
                    float tn[4]; bool h[4]; int idx[4]; int nh = 0;
                    for (int k = 0; k < nd.n; ++k) { h[k] = box_hit(nd.mn[k], nd.mx[k], o, inv, tbest, tn[k]); if (h[k]) idx[nh++] = k; }
                    std::sort(idx, idx + nh, [&](int a, int b) { return tn[a] < tn[b]; });
                    // leaves first (they can only shrink tbest), then inner children nearest first: the nearest continues, the rest are pushed far-first
                    for (int q = 0; q < nh; ++q) if (nd.child[idx[q]] < 0) tcount += leaf_test(T, nd.child[idx[q]], o, d, tbest, best);
                    int next = NONE;
                    for (int q = nh - 1; q >= 0; --q) {
                        const int k = idx[q];
                        if (nd.child[k] < 0 || tn[k] > tbest) continue;
                        if (next != NONE) { stack[sp] = next; stack_t[sp] = 0; ++sp; ++s4.pushes; s4.stack_max = std::max<long long>(s4.stack_max, sp); }
                        next = nd.child[k];
                    }
                    (void)stack_t;
                    if (next == NONE && sp > 0) next = stack[--sp];
                    cur = next;
                    s4.tris += tcount;
                    tri_per_iter[l].push_back(tcount);
                }
                it_max = std::max(it_max, it);
                ++s4.rays; s4.hits += best >= 0; depth4[r] = tbest;
            }
            s4.wave_iters += it_max; ++s4.waves;
            for (long long k = 0; k < it_max; ++k) { int m = 0; for (int l = 0; l < 64; ++l) if (k < (long long)tri_per_iter[l].size()) m = std::max(m, tri_per_iter[l][k]); s4.wave_tri_slots += m; }
        }
        long long diff = 0;
        for (int r = 0; r < nR; ++r) diff += depth2[r] != depth4[r];
        auto rep = [&](const char* nm, const Stat& s) {
            printf("  %-8s per ray: %.1f node fetches, %.1f triangle tests, hit fraction %.3f; per wave: %.1f loop iterations (slowest lane), %.1f triangle-test slots; "
                   "lane utilisation of node steps %.2f; pushes per ray %.1f, deepest stack %lld\n", nm, (double)s.nodes / s.rays, (double)s.tris / s.rays, (double)s.hits / s.rays,
                   (double)s.wave_iters / s.waves, (double)s.wave_tri_slots / s.waves, (double)s.nodes / (64.0 * s.wave_iters), (double)s.pushes / s.rays, s.stack_max);
        };
        rep("binary", s2); rep("4-wide", s4);
        printf("  rays whose depth differs between the two traversals: %lld of %d\n", diff, nR);
    }
    return 0;
}
