// bvh.hip -- triangle-mesh closest-hit ray tracer for Stage II (gfx950).  Replaces the un-vendored third-party CUDA extension
// `_raytracing` (ashawkey/raytracing) that the reference reaches through raytracing/raytracer.py:19 (create_raytracer) and :49
// (impl.trace): closest hit of N rays against a static mesh, outputs positions [N,3], geometric face normals [N,3] (winding
// order, unnormalised sign convention: cross(v1-v0, v2-v0), normalised here), depth [N]; a miss reports depth = 10 (the
// contract NeROMaterialRenderer.trace relies on: `miss_mask = depth >= 10`, network/renderer.py:727).
//
// Build: host C++, top-down median split on the longest centroid axis (std::nth_element), <= 4 triangles per leaf, nodes in
// DFS order.  A node is 64 bytes and carries BOTH children's boxes, so one 64-byte fetch decides two subtrees.
// Traversal: one ray per lane, closest-first descent; the BVH of a 1 M-triangle mesh is ~50 MB and lives in L2 / Infinity Cache.
// Two kernels walk the same tree in the same order with the same arithmetic per ray (bit-identical results):
//   * trace_overlap_kernel (default): a step requests the next node AND all triangles of the leaves it reached before it tests
//     anything, and keeps the deferred-subtree stack in LDS -- one memory round trip per step instead of up to six (measurements in
//     the comment above the kernel);
//   * trace_kernel: private (scratch) stack, one triangle after the other -- for trees deeper than the LDS stack, and the kernel the
//     default is tested against bit for bit (nero_bvh_set_traversal).
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cmath>
#include <cstring>
#include <vector>
#include "../../include/nero_hip.h"
#include "common.h"

namespace {

constexpr float MISS_DEPTH = 10.0f;
constexpr int NONE = -(1 << 30);          // "no node" sentinel (leaf references are > -2^28)

struct Node {                 // 64 bytes
    float lmin[3], lmax[3], rmin[3], rmax[3];
    int left, right;          // >= 0: node index; < 0: leaf, -(start*8 + count) - 1
    int pad[2];
};
struct Tri { float v0[3], e1[3], e2[3], pad[3]; };   // 48 bytes, leaf order

struct Bvh {
    Node* d_nodes = nullptr;
    Tri* d_tris = nullptr;
    int n_nodes = 0, n_tris = 0;
};

struct Builder {
    const float* V; const int* F;
    std::vector<int> order;
    std::vector<float> cen;          // centroids [nT,3]
    std::vector<float> bmin, bmax;   // per-triangle boxes
    std::vector<Node> nodes;
    int max_depth = 0;               // deepest inner node (root = 1): the traversal stack holds at most one entry per level

    void tri_box(int lo, int hi, float* mn, float* mx) const {
        for (int a = 0; a < 3; ++a) { mn[a] = 1e30f; mx[a] = -1e30f; }
        for (int i = lo; i < hi; ++i) {
            const int t = order[i];
            for (int a = 0; a < 3; ++a) { mn[a] = std::min(mn[a], bmin[t * 3 + a]); mx[a] = std::max(mx[a], bmax[t * 3 + a]); }
        }
    }
    // returns child reference for triangles [lo,hi)
    int build(int lo, int hi, int depth = 1) {
        const int n = hi - lo;
        if (n <= 4) return -(lo * 8 + n) - 1;
        max_depth = std::max(max_depth, depth);
        float cmn[3] = {1e30f, 1e30f, 1e30f}, cmx[3] = {-1e30f, -1e30f, -1e30f};
        for (int i = lo; i < hi; ++i)
            for (int a = 0; a < 3; ++a) { cmn[a] = std::min(cmn[a], cen[order[i] * 3 + a]); cmx[a] = std::max(cmx[a], cen[order[i] * 3 + a]); }
        int axis = 0;
        if (cmx[1] - cmn[1] > cmx[axis] - cmn[axis]) axis = 1;
        if (cmx[2] - cmn[2] > cmx[axis] - cmn[axis]) axis = 2;
        const int mid = lo + n / 2;
        std::nth_element(order.begin() + lo, order.begin() + mid, order.begin() + hi,
                         [&](int a, int b) { return cen[a * 3 + axis] < cen[b * 3 + axis]; });
        const int me = (int)nodes.size();
        nodes.emplace_back();
        Node nd;
        std::memset(&nd, 0, sizeof(nd));
        tri_box(lo, mid, nd.lmin, nd.lmax);
        tri_box(mid, hi, nd.rmin, nd.rmax);
        const int l = build(lo, mid, depth + 1);
        const int r = build(mid, hi, depth + 1);
        nd.left = l; nd.right = r;
        nodes[me] = nd;
        return me;
    }
};

__device__ __forceinline__ bool box_hit(const float* mn, const float* mx, const float* o, const float* inv, float tbest, float& tn) {
    float t0 = 0.f, t1 = tbest;
#pragma unroll
    for (int a = 0; a < 3; ++a) {
        float ta = (mn[a] - o[a]) * inv[a], tb = (mx[a] - o[a]) * inv[a];
        const float lo = fminf(ta, tb), hi = fmaxf(ta, tb);
        t0 = fmaxf(t0, lo);
        t1 = fminf(t1, hi);
    }
    tn = t0;
    return t0 <= t1;
}

// The arithmetic of a ray is the same in every traversal kernel below, operation by operation: contraction is switched off and every
// fused multiply-add is written out, so that the kernels agree bit for bit whatever hipcc makes of the code around the expressions.
struct TriQ { float4 a, b, c; };           // the 48 bytes of a Tri: v0 e1 | e1 e2 | e2 pad
__device__ __forceinline__ float cross_c(float a, float b, float c, float d) {     // a b - c d
#pragma clang fp contract(off)
    return fmaf(a, b, -(c * d));
}
__device__ __forceinline__ float dot3(float ax, float ay, float az, float bx, float by, float bz) {
#pragma clang fp contract(off)
    return fmaf(az, bz, fmaf(ay, by, ax * bx));
}
__device__ __forceinline__ void tri_test(const TriQ& q, int index, const float* o, const float* d, float& tbest, int& best) {
#pragma clang fp contract(off)
    const float v0[3] = {q.a.x, q.a.y, q.a.z}, e1[3] = {q.a.w, q.b.x, q.b.y}, e2[3] = {q.b.z, q.b.w, q.c.x};
    const float px = cross_c(d[1], e2[2], d[2], e2[1]), py = cross_c(d[2], e2[0], d[0], e2[2]), pz = cross_c(d[0], e2[1], d[1], e2[0]);
    const float det = dot3(e1[0], e1[1], e1[2], px, py, pz);
    if (fabsf(det) < 1e-20f) return;
    const float inv = 1.0f / det;
    const float tx = o[0] - v0[0], ty = o[1] - v0[1], tz = o[2] - v0[2];
    const float u = dot3(tx, ty, tz, px, py, pz) * inv;
    if (u < 0.f || u > 1.f) return;
    const float qx = cross_c(ty, e1[2], tz, e1[1]), qy = cross_c(tz, e1[0], tx, e1[2]), qz = cross_c(tx, e1[1], ty, e1[0]);
    const float v = dot3(d[0], d[1], d[2], qx, qy, qz) * inv;
    if (v < 0.f || u + v > 1.f) return;
    const float tt = dot3(e2[0], e2[1], e2[2], qx, qy, qz) * inv;
    if (tt > 0.f && tt < tbest) { tbest = tt; best = index; }
}
__device__ __forceinline__ TriQ load_tri(const Tri* __restrict__ tris, int i) {
    const float4* p = reinterpret_cast<const float4*>(tris + i);
    TriQ q;
    q.a = p[0]; q.b = p[1]; q.c = p[2];
    return q;
}
// the triangles of a leaf one after the other (low register use: trace_kernel)
__device__ __forceinline__ void leaf_test(const Tri* __restrict__ tris, int ref, const float* o, const float* d, float& tbest, int& best) {
    const int code = -ref - 1;
    const int start = code >> 3, count = code & 7;
    for (int i = 0; i < count; ++i) tri_test(load_tri(tris, start + i), start + i, o, d, tbest, best);
}
__device__ __forceinline__ void write_hit(const Tri* __restrict__ tris, int r, const float* o, const float* d, float tbest, int best,
                                          float* __restrict__ pos, float* __restrict__ nrm, float* __restrict__ depth) {
#pragma clang fp contract(off)
    if (best >= 0) {
        const TriQ q = load_tri(tris, best);
        const float e1[3] = {q.a.w, q.b.x, q.b.y}, e2[3] = {q.b.z, q.b.w, q.c.x};
        const float nx = cross_c(e1[1], e2[2], e1[2], e2[1]), ny = cross_c(e1[2], e2[0], e1[0], e2[2]), nz = cross_c(e1[0], e2[1], e1[1], e2[0]);
        const float nn = fmaxf(sqrtf(dot3(nx, ny, nz, nx, ny, nz)), 1e-30f);
        nrm[r * 3] = nx / nn; nrm[r * 3 + 1] = ny / nn; nrm[r * 3 + 2] = nz / nn;
    } else {
        nrm[r * 3] = 0.f; nrm[r * 3 + 1] = 0.f; nrm[r * 3 + 2] = 0.f;
    }
#pragma unroll
    for (int a = 0; a < 3; ++a) pos[r * 3 + a] = fmaf(tbest, d[a], o[a]);
    depth[r] = tbest;
}

__global__ __launch_bounds__(256) void trace_kernel(const Node* __restrict__ nodes, const Tri* __restrict__ tris, int root,
                                                    const float* __restrict__ ro, const float* __restrict__ rd, int n,
                                                    float* __restrict__ pos, float* __restrict__ nrm, float* __restrict__ depth,
                                                    const unsigned char* __restrict__ skip) {
    const int r = blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= n) return;
    const float o[3] = {ro[r * 3], ro[r * 3 + 1], ro[r * 3 + 2]};
    const float d[3] = {rd[r * 3], rd[r * 3 + 1], rd[r * 3 + 2]};
    float inv[3];
#pragma unroll
    for (int a = 0; a < 3; ++a) inv[a] = 1.0f / (fabsf(d[a]) > 1e-20f ? d[a] : (d[a] < 0.f ? -1e-20f : 1e-20f));
    float tbest = MISS_DEPTH;
    int best = -1;
    int stack[64];
    int sp = 0;
    int cur = (skip != nullptr && skip[r] != 0) ? NONE : root;          // a skipped ray reports a miss without a single node visit
    if (cur != NONE && cur < 0) { leaf_test(tris, cur, o, d, tbest, best); cur = NONE; }
    while (cur != NONE) {
        const Node nd = nodes[cur];
        float tl, tr;
        const bool hl = box_hit(nd.lmin, nd.lmax, o, inv, tbest, tl);
        const bool hr = box_hit(nd.rmin, nd.rmax, o, inv, tbest, tr);
        int next = NONE;
        int first = nd.left, second = nd.right;
        bool hf = hl, hs = hr;
        if (hl && hr && tr < tl) { first = nd.right; second = nd.left; }
        if (!hl) { first = nd.right; hf = hr; hs = false; }
        if (hf) {
            if (first < 0) leaf_test(tris, first, o, d, tbest, best); else next = first;
        }
        if (hs) {
            if (second < 0) leaf_test(tris, second, o, d, tbest, best);
            else if (next == NONE) next = second;
            else if (sp < 64) stack[sp++] = second;
        }
        if (next == NONE && sp > 0) next = stack[--sp];
        cur = next;
    }
    write_hit(tris, r, o, d, tbest, best, pos, nrm, depth);
}

// ---- one ray per thread, memory latencies overlapped --------------------------------------------------------------------------
// trace_kernel waits for memory several times per step: the node, then the triangles of a leaf ONE AFTER THE OTHER (the loop tests a
// triangle before it asks for the next), and a pop from the scratch stack before the node it names can be requested.  The counters
// say that is what the time is (scripts/prof_trace.sh, 1 M secondary rays: 76 % of the wave cycles parked in s_waitcnt, VALU busy
// 40 %, texture addresser busy 47 %, 3.5 waves per SIMD on average; 4.8 k cycles per step).  Here a step asks for everything at once:
// the NEXT node (its index is known as soon as the boxes are tested -- the leaf tests only shrink tbest) and all triangles of the
// leaves, then tests; the stack is in LDS.  Same visit order and the same arithmetic per ray as trace_kernel.
struct NodeQ { float4 a, b, c, d; };       // the 64 bytes of a Node: lmin lmax | rmin rmax | left right pad pad
__device__ __forceinline__ NodeQ load_node(const Node* __restrict__ nodes, int i) {
    const float4* p = reinterpret_cast<const float4*>(nodes + i);
    NodeQ q;
    q.a = p[0]; q.b = p[1]; q.c = p[2]; q.d = p[3];
    return q;
}
__device__ __forceinline__ void leaf_test_batched(const Tri* __restrict__ tris, int ref, const float* o, const float* d, float& tbest, int& best) {
    const int code = -ref - 1;
    const int start = code >> 3, count = code & 7;            // <= 4 (Builder::build)
    const float4* p = reinterpret_cast<const float4*>(tris + start);
    TriQ q[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {          // unconditional (a short leaf re-reads its first triangle): predicated loads come out of hipcc with a
        const int j = i < count ? i : 0;   // wait inside every predicated block, i.e. one triangle after the other again
        q[i].a = p[3 * j]; q[i].b = p[3 * j + 1]; q[i].c = p[3 * j + 2];
    }
#pragma unroll
    for (int i = 0; i < 4; ++i)
        if (i < count) tri_test(q[i], start + i, o, d, tbest, best);
}

// one wavefront per workgroup: the workgroups of a launch retire at very different times (secondary rays), and small ones refill the
// CUs sooner (1 M / 3.1 M secondary rays: 0.52 / 1.22 ms with 64 threads, 0.53 / 1.26 with 128, 0.56 / 1.38 with 256; coherent camera
// rays prefer 128: 0.30 against 0.34 ms per 1 M -- the training step's rays decide).  Measured and dropped: both leaves of a step
// requested together (116 VGPRs: +0.08 ms per 1 M rays); persistent wavefronts that refill idle lanes from a ray counter (42 % of the
// lanes are busy in a step, scripts/probe/trace_stats.cpp -- but the refilled wavefronts execute as many instructions, every step then
// has some lane in a leaf, on fewer resident waves: 0.92 ms); a 4-WIDE TREE (128-byte nodes collapsed from the binary tree, the four
// boxes of a node tested and sorted per step, hit leaves queued one per step; its visit logic checked on the CPU against this kernel's:
// same closest triangle on every ray) -- half the dependent steps per ray (36 -> 20), but 101 VGPRs and twice the box work per step:
// 0.57 ms per 1 M synthetic rays against 0.51, 0.83 against 0.73 on the rays of a training step.
struct ChunkOrder { int n; unsigned char c[32]; };            // n = 0: no explicit order
constexpr int PL_THREADS = 64;
constexpr int PL_STACK = 24;               // LDS stack entries per ray (6 KB per workgroup); deeper trees take trace_kernel

// LAUNCH ORDER (round 5, nero_bvh_trace_grouped): the secondary rays of Stage II come as [point][direction] with the cosine-weighted
// diffuse directions first and the GGX specular ones behind them.  Every ray that points below the geometric surface -- it crosses the
// inside of the mesh to its far side: hundreds of dependent node steps -- is a specular one, so the 64-ray chunks of a point fall into
// light ones (diffuse: above the horizon, most of them miss) and heavy ones, and in ray order the launch ends with whatever heavy chunks the
// last points own: 2.1 resident waves per SIMD on average (tracer).  With `gchunks` > 0 workgroup b takes the HEAVY chunks of
// all groups first ([heavy0, gchunks) of every group of gchunks chunks), then the light ones: the tail of the launch is made of short rays.
// Same rays, same arithmetic, same outputs at the same addresses -- only the order in which workgroups start changes.
__global__ __launch_bounds__(PL_THREADS) void trace_overlap_kernel(const Node* __restrict__ nodes, const Tri* __restrict__ tris, int root,
                                                                   const float* __restrict__ ro, const float* __restrict__ rd, int n,
                                                                   float* __restrict__ pos, float* __restrict__ nrm, float* __restrict__ depth,
                                                                   int gchunks, int heavy0, int n_groups, const unsigned char* __restrict__ skip,
                                                                   ChunkOrder ord) {
    __shared__ int lds_stack[PL_STACK * PL_THREADS];
    int* const st = lds_stack + threadIdx.x;                                   // entry s of this lane: st[s * PL_THREADS]
    int chunk = blockIdx.x;
    if (ord.n > 0) {                                     // phase p = all groups' chunk ord.c[p]: the workgroups of a phase start together
        if (chunk < n_groups * ord.n) chunk = (chunk % n_groups) * ord.n + ord.c[chunk / n_groups];
    } else if (gchunks > 0 && chunk < n_groups * gchunks) {
        const int nh = gchunks - heavy0, heavy_total = n_groups * nh;
        if (chunk < heavy_total) chunk = (chunk / nh) * gchunks + heavy0 + chunk % nh;
        else { const int b = chunk - heavy_total; chunk = (b / heavy0) * gchunks + b % heavy0; }
    }
    const int r = chunk * blockDim.x + threadIdx.x;
    if (r >= n) return;
    const float o[3] = {ro[r * 3], ro[r * 3 + 1], ro[r * 3 + 2]};
    const float d[3] = {rd[r * 3], rd[r * 3 + 1], rd[r * 3 + 2]};
    float inv[3];
#pragma unroll
    for (int a = 0; a < 3; ++a) inv[a] = 1.0f / (fabsf(d[a]) > 1e-20f ? d[a] : (d[a] < 0.f ? -1e-20f : 1e-20f));
    float tbest = MISS_DEPTH;
    int best = -1, sp = 0, cur = (skip != nullptr && skip[r] != 0) ? NONE : root;      // (a skipped ray: a miss without a node visit)
    NodeQ nd = {};
    if (cur == NONE) {}
    else if (cur < 0) { leaf_test(tris, cur, o, d, tbest, best); cur = NONE; }
    else nd = load_node(nodes, cur);
    // ONE leaf section per step: the second leaf of a node is tested at the start of the lane's next step, before its next node -- the
    // same order of visits, but a wavefront executes the triangle code once per step instead of twice (0.57 -> 0.52 ms per 1 M rays)
    int pend = NONE;
    while (cur != NONE || pend != NONE) {
        int leaf = pend;
        pend = NONE;
        if (leaf == NONE) {
            const float lmin[3] = {nd.a.x, nd.a.y, nd.a.z}, lmax[3] = {nd.a.w, nd.b.x, nd.b.y};
            const float rmin[3] = {nd.b.z, nd.b.w, nd.c.x}, rmax[3] = {nd.c.y, nd.c.z, nd.c.w};
            const int left = __float_as_int(nd.d.x), right = __float_as_int(nd.d.y);
            float tl, tr;
            const bool hl = box_hit(lmin, lmax, o, inv, tbest, tl);
            const bool hr = box_hit(rmin, rmax, o, inv, tbest, tr);
            int next = NONE, leaf_a = NONE, leaf_b = NONE;
            int first = left, second = right;
            bool hf = hl, hs = hr;
            if (hl && hr && tr < tl) { first = right; second = left; }
            if (!hl) { first = right; hf = hr; hs = false; }
            if (hf) {
                if (first < 0) leaf_a = first; else next = first;
            }
            if (hs) {
                if (second < 0) leaf_b = second;
                else if (next == NONE) next = second;
                else if (sp < PL_STACK) st[(sp++) * PL_THREADS] = second;
            }
            if (next == NONE && sp > 0) next = st[(--sp) * PL_THREADS];
            nd = load_node(nodes, next != NONE ? next : 0);                    // in flight while the triangles are fetched and tested.  Unconditional (a
                                                                               // finished lane re-reads node 0): behind `if (next != NONE)` hipcc reused a padding register of
                                                                               // the load as scratch and waited for the node before it requested the triangles (0.77 -> 0.73 ms
                                                                               // on the rays of a training step)
            cur = next;
            if (leaf_a != NONE) { leaf = leaf_a; pend = leaf_b; } else leaf = leaf_b;
        }
        if (leaf != NONE) leaf_test_batched(tris, leaf, o, d, tbest, best);
    }
    write_hit(tris, r, o, d, tbest, best, pos, nrm, depth);
}

struct Handle { Bvh b; int root; int max_depth; int mode; };

}  // namespace

extern "C" {

// vertices [nV,3] float32 and triangles [nT,3] int32 are HOST arrays (raytracing/raytracer.py:8-19 passes numpy arrays)
int nero_bvh_create(const float* verts, int nV, const int* tris, int nT, void** handle) {
    if (!verts || !tris || !handle || nT < 1 || nV < 3) return nero_fail(NERO_ERR_ARG, "nero_bvh_create: bad argument");
    Builder B;
    B.V = verts; B.F = tris;
    B.order.resize(nT); B.cen.resize((size_t)nT * 3); B.bmin.resize((size_t)nT * 3); B.bmax.resize((size_t)nT * 3);
    for (int t = 0; t < nT; ++t) {
        B.order[t] = t;
        for (int a = 0; a < 3; ++a) {
            float mn = 1e30f, mx = -1e30f, c = 0.f;
            for (int k = 0; k < 3; ++k) {
                const int vi = tris[t * 3 + k];
                if (vi < 0 || vi >= nV) return nero_fail(NERO_ERR_ARG, "nero_bvh_create: vertex index out of range");
                const float x = verts[(size_t)vi * 3 + a];
                mn = std::min(mn, x); mx = std::max(mx, x); c += x;
            }
            B.bmin[t * 3 + a] = mn; B.bmax[t * 3 + a] = mx; B.cen[t * 3 + a] = c / 3.f;
        }
    }
    B.nodes.reserve((size_t)nT / 2 + 16);
    const int root = B.build(0, nT);
    // closest-first descent pushes at most one deferred sibling per level: the 64-entry private stack of trace_kernel must cover
    // the tree depth, otherwise a far subtree would be dropped silently (median split: depth = ceil(log2(nT / 4)) + 1 <= 30)
    if (B.max_depth > 64) return nero_fail(NERO_ERR_UNSUPPORTED, "nero_bvh_create: BVH deeper than the 64-entry traversal stack");
    std::vector<Tri> T((size_t)nT);
    for (int i = 0; i < nT; ++i) {
        const int t = B.order[i];
        const float* a = verts + (size_t)tris[t * 3] * 3;
        const float* b = verts + (size_t)tris[t * 3 + 1] * 3;
        const float* c = verts + (size_t)tris[t * 3 + 2] * 3;
        for (int k = 0; k < 3; ++k) { T[i].v0[k] = a[k]; T[i].e1[k] = b[k] - a[k]; T[i].e2[k] = c[k] - a[k]; T[i].pad[k] = 0.f; }
    }
    Handle* h = new Handle();
    h->root = root;
    h->max_depth = B.max_depth;
    h->mode = B.max_depth <= PL_STACK ? 1 : 0;             // a deferred sibling per level: deeper trees take the private-stack kernel
    h->b.n_nodes = (int)B.nodes.size();
    h->b.n_tris = nT;
    const size_t nb = std::max<size_t>(1, B.nodes.size()) * sizeof(Node);
    if (hipMalloc(&h->b.d_nodes, nb) != hipSuccess || hipMalloc(&h->b.d_tris, (size_t)nT * sizeof(Tri)) != hipSuccess) {
        (void)hipFree(h->b.d_nodes);
        (void)hipFree(h->b.d_tris);
        delete h;
        return nero_fail(NERO_ERR_LAUNCH, "nero_bvh_create: hipMalloc failed");
    }
    hipError_t e0 = hipSuccess;
    if (!B.nodes.empty()) e0 = hipMemcpy(h->b.d_nodes, B.nodes.data(), B.nodes.size() * sizeof(Node), hipMemcpyHostToDevice);
    const hipError_t e1 = hipMemcpy(h->b.d_tris, T.data(), (size_t)nT * sizeof(Tri), hipMemcpyHostToDevice);
    if (e0 != hipSuccess || e1 != hipSuccess) {
        (void)hipFree(h->b.d_nodes);
        (void)hipFree(h->b.d_tris);
        delete h;
        return nero_fail(NERO_ERR_LAUNCH, "nero_bvh_create: hipMemcpy of the BVH failed");
    }
    *handle = h;
    return NERO_OK;
}

int nero_bvh_trace(void* handle, const float* rays_o, const float* rays_d, int n, float* positions, float* face_normals, float* depth,
                   void* stream) {
    if (!handle || !rays_o || !rays_d || !positions || !face_normals || !depth) return nero_fail(NERO_ERR_ARG, "nero_bvh_trace: bad argument");
    if (n == 0) return NERO_OK;
    Handle* h = (Handle*)handle;
    if (h->mode == 0) {
        hipLaunchKernelGGL(trace_kernel, dim3((n + 255) / 256), dim3(256), 0, (hipStream_t)stream, h->b.d_nodes, h->b.d_tris, h->root,
                           rays_o, rays_d, n, positions, face_normals, depth, nullptr);
        return nero_check_launch("nero_bvh_trace");
    }
    hipLaunchKernelGGL(trace_overlap_kernel, dim3((n + PL_THREADS - 1) / PL_THREADS), dim3(PL_THREADS), 0, (hipStream_t)stream, h->b.d_nodes, h->b.d_tris,
                       h->root, rays_o, rays_d, n, positions, face_normals, depth, 0, 0, 0, nullptr, ChunkOrder{});
    return nero_check_launch("nero_bvh_trace");
}

// nero_bvh_trace for a caller that knows some rays' results will not be used: skip [n] bytes, non-zero = do not traverse -- the ray is
// reported as a miss (depth 10, zero normal) at the cost of one byte read.  Stage II (round 6): the GGX-sampled directions below the shading
// horizon carry an estimator weight of exactly zero (nero_mc_dead_rays) and are the longest rays of the launch (they cross the inside of
// the mesh).  Every other ray's outputs are those of nero_bvh_trace bit for bit.  skip = NULL: nero_bvh_trace.
int nero_bvh_trace_masked(void* handle, const float* rays_o, const float* rays_d, int n, const unsigned char* skip, float* positions,
                          float* face_normals, float* depth, void* stream) {
    if (!handle || !rays_o || !rays_d || !positions || !face_normals || !depth) return nero_fail(NERO_ERR_ARG, "nero_bvh_trace_masked: bad argument");
    if (n == 0) return NERO_OK;
    Handle* h = (Handle*)handle;
    if (h->mode == 0) {
        hipLaunchKernelGGL(trace_kernel, dim3((n + 255) / 256), dim3(256), 0, (hipStream_t)stream, h->b.d_nodes, h->b.d_tris, h->root,
                           rays_o, rays_d, n, positions, face_normals, depth, skip);
        return nero_check_launch("nero_bvh_trace_masked");
    }
    hipLaunchKernelGGL(trace_overlap_kernel, dim3((n + PL_THREADS - 1) / PL_THREADS), dim3(PL_THREADS), 0, (hipStream_t)stream, h->b.d_nodes, h->b.d_tris,
                       h->root, rays_o, rays_d, n, positions, face_normals, depth, 0, 0, 0, skip, ChunkOrder{});
    return nero_check_launch("nero_bvh_trace_masked");
}

// nero_bvh_trace with a launch-order hint for rays that come in groups of `group` (Stage II: the D = Dd + Ds directions of a surface
// point) whose entries [heavy_from, group) are the expensive ones (the specular directions): those chunks are started first.  Outputs
// are identical to nero_bvh_trace's.  group and heavy_from must be multiples of 64 (the chunk size) with 0 < heavy_from < group, and n a
// multiple of group; anything else takes the natural order.
int nero_bvh_trace_grouped(void* handle, const float* rays_o, const float* rays_d, int n, float* positions, float* face_normals, float* depth,
                           int group, int heavy_from, void* stream) {
    if (!handle || !rays_o || !rays_d || !positions || !face_normals || !depth) return nero_fail(NERO_ERR_ARG, "nero_bvh_trace_grouped: bad argument");
    Handle* h = (Handle*)handle;
    const bool ok = h->mode != 0 && group > 0 && heavy_from > 0 && heavy_from < group && group % PL_THREADS == 0 && heavy_from % PL_THREADS == 0 &&
                    n > 0 && n % group == 0;
    if (!ok) return nero_bvh_trace(handle, rays_o, rays_d, n, positions, face_normals, depth, stream);
    hipLaunchKernelGGL(trace_overlap_kernel, dim3(n / PL_THREADS), dim3(PL_THREADS), 0, (hipStream_t)stream, h->b.d_nodes, h->b.d_tris, h->root,
                       rays_o, rays_d, n, positions, face_normals, depth, group / PL_THREADS, heavy_from / PL_THREADS, n / group, nullptr, ChunkOrder{});
    return nero_check_launch("nero_bvh_trace_grouped");
}

// nero_bvh_trace_masked with an explicit LAUNCH ORDER (round 6).  The rays come in groups of n_order * 64 (Stage II: the D directions of a
// surface point); phase p of the launch holds chunk order[p] (64 rays) of EVERY group, so the workgroups of the chunks named first start first.
// Stage II passes the chunks in descending order: both direction tables run from the pole of their lobe outwards (field.py:741-749), the
// later entries of a table are the grazing directions -- the long traversals -- and a launch that ends with them runs its last
// hundreds of microseconds on a handful of resident waves (tracer 0.69 -> 0.57 ms per C4 step).  Same rays, same arithmetic, same outputs at the
// same addresses.  order: a permutation of 0 .. n_order-1, n_order <= 32, n % (64 n_order) == 0, LDS-stack kernel only; anything else: natural order.
int nero_bvh_trace_ordered(void* handle, const float* rays_o, const float* rays_d, int n, const unsigned char* skip, const int* order, int n_order,
                           float* positions, float* face_normals, float* depth, void* stream) {
    if (!handle || !rays_o || !rays_d || !positions || !face_normals || !depth) return nero_fail(NERO_ERR_ARG, "nero_bvh_trace_ordered: bad argument");
    Handle* h = (Handle*)handle;
    bool ok = h->mode != 0 && order && n_order > 0 && n_order <= 32 && n > 0 && n % (n_order * PL_THREADS) == 0;
    ChunkOrder ord{};
    unsigned seen = 0;
    for (int i = 0; ok && i < n_order; ++i) {
        if (order[i] < 0 || order[i] >= n_order || (seen >> order[i] & 1u)) ok = false;
        else { seen |= 1u << order[i]; ord.c[i] = (unsigned char)order[i]; }
    }
    if (!ok) return nero_bvh_trace_masked(handle, rays_o, rays_d, n, skip, positions, face_normals, depth, stream);
    ord.n = n_order;
    hipLaunchKernelGGL(trace_overlap_kernel, dim3(n / PL_THREADS), dim3(PL_THREADS), 0, (hipStream_t)stream, h->b.d_nodes, h->b.d_tris, h->root,
                       rays_o, rays_d, n, positions, face_normals, depth, 0, 0, n / (n_order * PL_THREADS), skip, ord);
    return nero_check_launch("nero_bvh_trace_ordered");
}

int nero_bvh_set_traversal(void* handle, int mode) {
    if (!handle || mode < 0 || mode > 1) return nero_fail(NERO_ERR_ARG, "nero_bvh_set_traversal: bad argument");
    Handle* h = (Handle*)handle;
    if (mode == 1 && h->max_depth > PL_STACK) return nero_fail(NERO_ERR_UNSUPPORTED, "nero_bvh_set_traversal: tree deeper than the LDS traversal stack");
    h->mode = mode;
    return NERO_OK;
}

int nero_bvh_destroy(void* handle) {
    if (!handle) return NERO_OK;
    Handle* h = (Handle*)handle;
    (void)hipFree(h->b.d_nodes);
    (void)hipFree(h->b.d_tris);
    delete h;
    return NERO_OK;
}

}  // extern "C"
