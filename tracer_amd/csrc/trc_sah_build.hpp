// trc_sah_build.hpp -- the reference's binned-SAH tree build on the device (included by trc_lbvh.hip, inside its
// unnamed namespace, after the kernels it shares: DLeaf, DTopo, ordered_of / float_of, the wave scan).
//
// What is built is BVH::buildTree / BVH::make of RT_Metal/Metal/BVH.hh:35-269 -- centroid bounds, widest axis, 10 buckets,
// cost_i = 1 + (n0 A0 + n1 A1) / A(centroid box), first minimum, partition by bucket <= split with the two-ended swap
// loop of BVH.hh:154-168, median split where the partition is one-sided, two leaves ordered by centroid -- so the tree is
// the one tracer_amd/host/bvh_builder.cpp builds on the host and oracle/oracle_sah.cpp restates, record for record (same
// post-order interior numbering, same leaf order inside every node).  The host recursion becomes:
//
//   large nodes (more than kFinishSpan leaves), one LEVEL per round of four launches over the leaf records, which are kept
//   in tree order (32 B each, moved when a partition moves them: every pass streams contiguous memory):
//     k_sah_bins     per task (<= kChunk consecutive records of one node): bucket counts and boxes in LDS, one set of
//                    atomics per task into the node's row; the task's bucket histogram is kept
//     k_sah_split    one lane per node: the nine costs, the split, the node's topology record; its children's rows, tasks
//                    and finish entries (one atomic per wavefront), a scan of the task histograms (how many records below
//                    the split precede each task)
//     k_sah_scatter  per task: the two-ended swap loop in closed form -- the k-th misplaced record from the left changes
//                    places with the k-th misplaced record from the right -- misplaced records go to a side array by rank;
//                    the children's centroid bounds are accumulated on the way
//     k_sah_apply    misplaced positions take their partner's record
//   small nodes: k_sah_finish, one wavefront per subtree of <= 64 leaves, one lane per leaf, the whole subtree in LDS.
//
// Integer / min-max work on 32-B records: streaming passes bound by HBM and launch latency, nothing for MFMA.
// Preconditions checked on the device: finite boxes with |coordinates| <= 1e37 (then a one-sided partition can only
// come from identical centroids, where the reference's stable sort leaves the order alone).

constexpr uint32_t kSahBuckets = 10;
constexpr uint32_t kFinishSpan = 64;        // subtrees of at most this many leaves are finished by one wavefront
constexpr uint32_t kChunk = 2048;           // records per task of the level passes
constexpr uint32_t kNoNode = 0xFFFFFFFFu;

struct SNode {                              // a node of the level being split; 88 dwords
    uint32_t first, last, qbase, task0;     // record range, first post-order slot of its subtree, first task
    uint32_t cb[6];                         // centroid bounds as ordered uints (min xyz, max xyz)
    uint32_t split, mid, fallback, n_below; // written by k_sah_split
    uint32_t child[2];                      // rows of the children in the next level's table, or kNoNode
    uint32_t child_t0[2];                   // their first tasks
    uint32_t count[kSahBuckets];
    uint32_t bb[kSahBuckets][6];            // bucket boxes as ordered uints
};
static_assert(sizeof(SNode) == 88 * 4 && offsetof(SNode, cb) == 16, "row layout");
struct SFinish { uint32_t first, last, qbase, depth; };      // depth of the subtree's root

struct SahBufs {
    DLeaf* elems; DLeaf* tmp;               // records in tree order; side array of the partition
    uint32_t* mv;                           // per position: 0 = stays, else (rank + 1) | side << 31
    SNode* nodes[2]; uint32_t* task_node[2];
    uint32_t* task_hist;
    SFinish* finish;
    uint32_t* counters;                     // [0] finish entries, [1] error bits, [2] depth of the deepest leaf, [2 + 2 * level ...] nodes / tasks of level >= 1
    uint32_t n;
};

__device__ __forceinline__ uint32_t sah_id_of(uint32_t q, uint32_t n) { return q + 2u == n ? 0u : q + 1u; }   // post-order slot -> interior index (root = 0)
__device__ __forceinline__ float sah_centroid1(float mn, float mx) { return mn + (mx - mn) / 2.0f; }           // AABB.hh:22-25
__device__ __forceinline__ uint32_t sah_widest(float dx, float dy, float dz) {                                   // AABB.hh:42-49
    if (dx > dy && dx > dz) return 0u;
    return dy > dz ? 1u : 2u;
}
__device__ __forceinline__ uint32_t sah_bucket(float c, float lo, float extent) {                                // BVH.hh:99-100
    const float rel = (c - lo) / extent;
    const float fb = (float)kSahBuckets * rel;
    const uint32_t b = (fb >= 0.0f) ? (uint32_t)fb : 0u;
    return b < kSahBuckets - 1u ? b : kSahBuckets - 1u;
}
__device__ __forceinline__ float sah_area(const float mn[3], const float mx[3]) {                               // AABB.hh:27-30
    const float dx = mx[0] - mn[0], dy = mx[1] - mn[1], dz = mx[2] - mn[2];
    return 2 * (dx * dy + dx * dz + dy * dz);
}

// row 0 of the first level's table: the root over all n records, identity bounds and bins
__global__ void __launch_bounds__(128) k_sah_root_row(SNode* nodes, uint32_t n) {
    uint32_t* w = (uint32_t*)&nodes[0];
    const uint32_t i = threadIdx.x;
    if (i >= 88u) return;
    uint32_t v = 0u;
    if (i == 1u) v = n;                                                     // first = 0, last = n, qbase = 0, task0 = 0
    else if (i >= 4u && i < 10u) v = (i - 4u) < 3u ? 0xFFFFFFFFu : 0u;      // cb
    else if (i >= 28u) v = ((i - 28u) % 6u) < 3u ? 0xFFFFFFFFu : 0u;        // bb
    w[i] = v;
}

// records in tree order start as the leaves in input order; _pad carries the leaf index.  bad bit 8: a box the build
// does not take (non-finite, or beyond 1e37).  root != nullptr: the centroid bounds of all records go to its row (wave
// reduction, LDS across the four wavefronts, 6 atomics per workgroup).
__global__ void __launch_bounds__(256) k_sah_init(const DLeaf* leaves, uint32_t n, DLeaf* elems, uint32_t* bad, SNode* root) {
    __shared__ uint32_t part[4][6];
    const uint32_t i = blockIdx.x * 256u + threadIdx.x;
    uint32_t lo[3] = {0xFFFFFFFFu, 0xFFFFFFFFu, 0xFFFFFFFFu}, hi[3] = {0u, 0u, 0u};
    if (i < n) {
        DLeaf d = leaves[i];
        d._pad = i;
        bool ok = true;
#pragma unroll
        for (int a = 0; a < 3; ++a) ok = ok && fabsf(d.mn[a]) <= 1e37f && fabsf(d.mx[a]) <= 1e37f;      // false for NaN too
        if (!ok) atomicOr(bad, 8u);
        elems[i] = d;
#pragma unroll
        for (int a = 0; a < 3; ++a) lo[a] = hi[a] = ordered_of(sah_centroid1(d.mn[a], d.mx[a]));
    }
    if (!root) return;
#pragma unroll
    for (int a = 0; a < 3; ++a) {
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) {
            lo[a] = min(lo[a], (uint32_t)__shfl_xor((int)lo[a], off, 64));
            hi[a] = max(hi[a], (uint32_t)__shfl_xor((int)hi[a], off, 64));
        }
    }
    const uint32_t lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
    if (lane == 0) {
#pragma unroll
        for (int a = 0; a < 3; ++a) { part[wave][a] = lo[a]; part[wave][3 + a] = hi[a]; }
    }
    __syncthreads();
    if (threadIdx.x < 3) {
        const uint32_t a = threadIdx.x;
        atomicMin(&root->cb[a], min(min(part[0][a], part[1][a]), min(part[2][a], part[3][a])));
        atomicMax(&root->cb[3 + a], max(max(part[0][3 + a], part[1][3 + a]), max(part[2][3 + a], part[3][3 + a])));
    }
}

__device__ __forceinline__ uint32_t wave_max_u32(uint32_t v) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v = max(v, (uint32_t)__shfl_xor((int)v, off, 64));
    return v;
}
struct SahAxis { uint32_t axis; float lo, extent; };
__device__ __forceinline__ SahAxis sah_axis_of(const uint32_t cb[6]) {
    float lo[3], hi[3];
#pragma unroll
    for (int a = 0; a < 3; ++a) { lo[a] = float_of(cb[a]); hi[a] = float_of(cb[3 + a]); }
    SahAxis r;
    r.axis = sah_widest(hi[0] - lo[0], hi[1] - lo[1], hi[2] - lo[2]);
    r.lo = r.axis == 0 ? lo[0] : (r.axis == 1 ? lo[1] : lo[2]);
    r.extent = (r.axis == 0 ? hi[0] : (r.axis == 1 ? hi[1] : hi[2])) - r.lo;
    return r;
}
__device__ __forceinline__ float sah_axis_centroid(const DLeaf& d, uint32_t axis) {
    return axis == 0 ? sah_centroid1(d.mn[0], d.mx[0]) : (axis == 1 ? sah_centroid1(d.mn[1], d.mx[1]) : sah_centroid1(d.mn[2], d.mx[2]));
}

// The LDS bins exist in kBinCopies copies (lane & 31 picks one, odd stride: the copies start in different banks): the
// records of a task mostly fall into one or two buckets, and 64 lanes on one LDS address are 64 serial atomics.
constexpr uint32_t kBinCopies = 32, kBinStride = 71;        // 70 words per copy: 10 counts, 10 x 6 bounds
__global__ void __launch_bounds__(256) k_sah_bins(const DLeaf* elems, SNode* nodes, const uint32_t* task_node, uint32_t* task_hist) {
    __shared__ uint32_t s_bin[kBinCopies * kBinStride];
    const uint32_t t = blockIdx.x, node = task_node[t];
    SNode& N = nodes[node];
    const uint32_t p0 = N.first + (t - N.task0) * kChunk, p1 = min(N.last, p0 + kChunk);
    const SahAxis ax = sah_axis_of(N.cb);
    for (uint32_t i = threadIdx.x; i < kBinCopies * kBinStride; i += 256u) {
        const uint32_t w = i % kBinStride;
        s_bin[i] = (w >= kSahBuckets && w < 70u && ((w - kSahBuckets) % 6u) < 3u) ? 0xFFFFFFFFu : 0u;
    }
    __syncthreads();
    uint32_t* mine = s_bin + (threadIdx.x & (kBinCopies - 1u)) * kBinStride;
    for (uint32_t p = p0 + threadIdx.x; p < p1; p += 256u) {
        const DLeaf d = elems[p];
        const uint32_t b = sah_bucket(sah_axis_centroid(d, ax.axis), ax.lo, ax.extent);
        atomicAdd(&mine[b], 1u);
        uint32_t* bb = mine + kSahBuckets + b * 6u;
#pragma unroll
        for (int a = 0; a < 3; ++a) {
            atomicMin(&bb[a], ordered_of(d.mn[a]));
            atomicMax(&bb[3 + a], ordered_of(d.mx[a]));
        }
    }
    __syncthreads();
    if (threadIdx.x < 70u) {
        const uint32_t w = threadIdx.x;
        const bool is_count = w < kSahBuckets, is_min = !is_count && ((w - kSahBuckets) % 6u) < 3u;
        uint32_t v = is_min ? 0xFFFFFFFFu : 0u;
        for (uint32_t c = 0; c < kBinCopies; ++c) {
            const uint32_t x = s_bin[c * kBinStride + w];
            v = is_count ? v + x : (is_min ? min(v, x) : max(v, x));
        }
        s_bin[w] = v;                                   // copy 0 becomes the task's total (each thread rewrites only the word it read last)
    }
    __syncthreads();
    if (threadIdx.x < kSahBuckets) {
        const uint32_t c = s_bin[threadIdx.x];
        task_hist[t * kSahBuckets + threadIdx.x] = c;
        if (c) atomicAdd(&N.count[threadIdx.x], c);
    } else if (threadIdx.x < 70u) {
        const uint32_t w = threadIdx.x - kSahBuckets, b = w / 6u, a = w % 6u;
        if (s_bin[b]) { if (a < 3u) atomicMin(&N.bb[b][a], s_bin[threadIdx.x]); else atomicMax(&N.bb[b][a], s_bin[threadIdx.x]); }
    }
}

// One LANE per node of the level (64 nodes per wavefront): the lane prices the nine splits of BVH.hh:112-140 from the node's
// bucket rows -- prefix boxes left to right, suffix boxes right to left: min / max are exact, so the order of the unions does
// not matter -- decides, and writes the node's topology record.  Rows, tasks and finish entries of the children are
// handed out with ONE atomic per wavefront and counter (a level of 15 000 nodes otherwise queues 45 000 atomics on three
// addresses); what needs many words per node -- identity bins of the children's rows, their task lists, the scan
// of a many-task node's histograms -- is left to the passes that have a workgroup per task (k_sah_scatter, k_sah_apply).
__global__ void __launch_bounds__(64) k_sah_split(SNode* nodes, uint32_t n_rows, SNode* next_nodes, uint32_t* next_counts /* [2]: nodes, tasks */, SFinish* finish,
                                                 uint32_t* counters, DTopo tp, uint32_t n, uint32_t level) {
    const uint32_t lane = threadIdx.x, row = blockIdx.x * 64u + lane;
    const bool has = row < n_rows;
    uint32_t first = 0, last = 0, qbase = 0, split = 0, mid = 0;
    uint32_t c_first[2] = {0, 0}, c_last[2] = {0, 0}, c_q[2] = {0, 0}, c_tasks[2] = {0, 0};
    uint32_t need_rows = 0, need_tasks = 0, need_finish = 0;
    bool c_row[2] = {false, false}, c_fin[2] = {false, false};
    if (has) {
        SNode& N = nodes[row];
        first = N.first; last = N.last; qbase = N.qbase;
        const uint32_t span = last - first;
        const SahAxis ax = sah_axis_of(N.cb);
        uint32_t n_below = 0;
        bool fallback = !(ax.extent > 0.0f);
        if (!fallback) {
            uint32_t count[kSahBuckets];
            float bmn[kSahBuckets][3], bmx[kSahBuckets][3];
#pragma unroll
            for (uint32_t b = 0; b < kSahBuckets; ++b) {
                count[b] = N.count[b];
#pragma unroll
                for (int a = 0; a < 3; ++a) {
                    bmn[b][a] = count[b] ? float_of(N.bb[b][a]) : FLT_MAX;
                    bmx[b][a] = count[b] ? float_of(N.bb[b][3 + a]) : -FLT_MAX;
                }
            }
            float smn[kSahBuckets][3], smx[kSahBuckets][3];         // [i]: buckets i .. 9
            int sc[kSahBuckets];
#pragma unroll
            for (int a = 0; a < 3; ++a) { smn[kSahBuckets - 1][a] = bmn[kSahBuckets - 1][a]; smx[kSahBuckets - 1][a] = bmx[kSahBuckets - 1][a]; }
            sc[kSahBuckets - 1] = (int)count[kSahBuckets - 1];
#pragma unroll
            for (int i = (int)kSahBuckets - 2; i >= 1; --i) {
#pragma unroll
                for (int a = 0; a < 3; ++a) { smn[i][a] = fminf(bmn[i][a], smn[i + 1][a]); smx[i][a] = fmaxf(bmx[i][a], smx[i + 1][a]); }
                sc[i] = sc[i + 1] + (int)count[i];
            }
            float cmn[3], cmx[3];
#pragma unroll
            for (int a = 0; a < 3; ++a) { cmn[a] = float_of(N.cb[a]); cmx[a] = float_of(N.cb[3 + a]); }
            const float denom = sah_area(cmn, cmx);
            float pmn[3] = {FLT_MAX, FLT_MAX, FLT_MAX}, pmx[3] = {-FLT_MAX, -FLT_MAX, -FLT_MAX};
            int pc = 0;
            float best = 0.0f;
#pragma unroll
            for (uint32_t i = 0; i + 1 < kSahBuckets; ++i) {
#pragma unroll
                for (int a = 0; a < 3; ++a) { pmn[a] = fminf(pmn[a], bmn[i][a]); pmx[a] = fmaxf(pmx[a], bmx[i][a]); }
                pc += (int)count[i];
                const float cost = 1 + ((float)pc * sah_area(pmn, pmx) + (float)sc[i + 1] * sah_area(smn[i + 1], smx[i + 1])) / denom;
                if (i == 0 || cost < best) { best = cost; split = i; n_below = (uint32_t)pc; }      // first minimum, BVH.hh:133-140
            }
            mid = first + n_below;
            fallback = mid <= first || mid >= last;
            if (fallback) atomicOr(&counters[1], 16u);          // cannot happen with a positive finite extent (header)
        }
        if (fallback) mid = first + span / 2;
        N.split = split; N.mid = mid; N.fallback = fallback ? 1u : 0u; N.n_below = n_below;
        // topology of this node; what its children need
        const uint32_t self = sah_id_of(qbase + span - 2u, n);
        uint32_t link[2];
#pragma unroll
        for (int side = 0; side < 2; ++side) {
            const uint32_t cf = side ? mid : first, cl = side ? last : mid, cs = cl - cf;
            const uint32_t cq = side ? qbase + (mid - first) - 1u : qbase;
            c_first[side] = cf; c_last[side] = cl; c_q[side] = cq;
            if (cs == 1u) { link[side] = kChildLeaf | cf; tp.parent_leaf[cf] = self; atomicMax(&counters[2], level + 1u); continue; }
            const uint32_t ci = sah_id_of(cq + cs - 2u, n);
            link[side] = ci; tp.parent_interior[ci] = self;
            if (cs <= kFinishSpan) { c_fin[side] = true; ++need_finish; }
            else { c_row[side] = true; c_tasks[side] = (cs + kChunk - 1) / kChunk; ++need_rows; need_tasks += c_tasks[side]; }
        }
        tp.child_l[self] = link[0]; tp.child_r[self] = link[1]; tp.axis[self] = ax.axis;
        if (self == 0u) tp.parent_interior[0] = 0u;
    }
    // one atomic per wavefront and counter
    const uint32_t inc_rows = wave_inclusive_scan(need_rows, lane), inc_tasks = wave_inclusive_scan(need_tasks, lane),
                   inc_fin = wave_inclusive_scan(need_finish, lane);
    uint32_t base_rows = 0, base_tasks = 0, base_fin = 0;
    if (lane == 63u) {
        if (inc_rows) { base_rows = atomicAdd(&next_counts[0], inc_rows); base_tasks = atomicAdd(&next_counts[1], inc_tasks); }
        if (inc_fin) base_fin = atomicAdd(&counters[0], inc_fin);
    }
    uint32_t my_row = (uint32_t)__shfl((int)base_rows, 63, 64) + inc_rows - need_rows;
    uint32_t my_task = (uint32_t)__shfl((int)base_tasks, 63, 64) + inc_tasks - need_tasks;
    uint32_t my_fin = (uint32_t)__shfl((int)base_fin, 63, 64) + inc_fin - need_finish;
    uint32_t child_row[2] = {kNoNode, kNoNode}, child_t0[2] = {0, 0};
#pragma unroll
    for (int side = 0; side < 2; ++side) {
        if (c_fin[side]) { SFinish f; f.first = c_first[side]; f.last = c_last[side]; f.qbase = c_q[side]; f.depth = level + 1u; finish[my_fin++] = f; }
        if (c_row[side]) {
            child_row[side] = my_row++; child_t0[side] = my_task; my_task += c_tasks[side];
            SNode& c = next_nodes[child_row[side]];        // header and identity centroid bounds (k_sah_scatter accumulates them);
            c.first = c_first[side]; c.last = c_last[side]; c.qbase = c_q[side]; c.task0 = child_t0[side];      // the bins are reset by k_sah_apply
#pragma unroll
            for (int a = 0; a < 6; ++a) c.cb[a] = a < 3 ? 0xFFFFFFFFu : 0u;
        }
    }
    if (has) { SNode& N = nodes[row]; N.child[0] = child_row[0]; N.child[1] = child_row[1]; N.child_t0[0] = child_t0[0]; N.child_t0[1] = child_t0[1]; }
}

// exclusive scan of a predicate over the 256 threads of a workgroup; wave_tot: 4 words of LDS
__device__ __forceinline__ uint32_t block_pred_scan256(bool pred, uint32_t* wave_tot, uint32_t& total) {
    const uint32_t lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
    const unsigned long long m = __ballot(pred);
    if (lane == 0) wave_tot[wave] = (uint32_t)__popcll(m);
    __syncthreads();
    uint32_t before = 0;
    for (uint32_t w = 0; w < wave; ++w) before += wave_tot[w];
    total = wave_tot[0] + wave_tot[1] + wave_tot[2] + wave_tot[3];
    __syncthreads();
    return before + (uint32_t)__popcll(m & ((1ull << lane) - 1ull));
}

__global__ void __launch_bounds__(256) k_sah_scatter(const DLeaf* elems, DLeaf* tmp, uint32_t* mv, const SNode* nodes, SNode* next_nodes,
                                                    const uint32_t* task_node, const uint32_t* task_hist, uint32_t* next_task_node) {
    __shared__ uint32_t wave_tot[4], s_prefix;
    __shared__ uint32_t s_cb[2][6];
    const uint32_t t = blockIdx.x, node = task_node[t];
    const SNode& N = nodes[node];
    const uint32_t first = N.first, last = N.last, mid = N.mid, split = N.split, n_below = N.n_below;
    const bool fallback = N.fallback != 0u;
    const uint32_t p0 = first + (t - N.task0) * kChunk, p1 = min(last, p0 + kChunk);
    const SahAxis ax = sah_axis_of(N.cb);
    if (threadIdx.x < 12u) s_cb[threadIdx.x / 6u][threadIdx.x % 6u] = (threadIdx.x % 6u) < 3u ? 0xFFFFFFFFu : 0u;
    uint32_t lo[2][3] = {{0xFFFFFFFFu, 0xFFFFFFFFu, 0xFFFFFFFFu}, {0xFFFFFFFFu, 0xFFFFFFFFu, 0xFFFFFFFFu}}, hi[2][3] = {{0u, 0u, 0u}, {0u, 0u, 0u}};
    // records below the split in the node's tasks before this one: the histograms k_sah_bins kept (a few KB from L2)
    const uint32_t k_task = t - N.task0, n_tasks = (last - first + kChunk - 1) / kChunk;
    if (threadIdx.x == 0) s_prefix = 0u;
    __syncthreads();
    if (!fallback && k_task) {
        uint32_t v = 0;
        for (uint32_t i = threadIdx.x; i < k_task * (split + 1u); i += 256u)
            v += task_hist[(size_t)(N.task0 + i / (split + 1u)) * kSahBuckets + i % (split + 1u)];
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) v += (uint32_t)__shfl_xor((int)v, off, 64);
        if ((threadIdx.x & 63u) == 0u && v) atomicAdd(&s_prefix, v);
    }
    // the children's task lists: entry k of [left's tasks, right's tasks] is written by task k, the one beyond by the last task
    if (threadIdx.x < 2u) {
        const uint32_t nl = N.child[0] != kNoNode ? (mid - first + kChunk - 1) / kChunk : 0u;
        const uint32_t nr = N.child[1] != kNoNode ? (last - mid + kChunk - 1) / kChunk : 0u;
        const uint32_t e = k_task + threadIdx.x * n_tasks;                 // thread 1: entry n_tasks + k_task, only k_task == 0 can have one
        if (threadIdx.x == 0u || k_task == 0u) {
            if (e < nl) next_task_node[N.child_t0[0] + e] = N.child[0];
            else if (e - nl < nr) next_task_node[N.child_t0[1] + (e - nl)] = N.child[1];
        }
    }
    __syncthreads();
    uint32_t running = s_prefix;
    for (uint32_t base = p0; base < p1; base += 256u) {
        const uint32_t p = base + threadIdx.x;
        const bool valid = p < p1;
        DLeaf d{};
        if (valid) d = elems[p];
        float c[3];
#pragma unroll
        for (int a = 0; a < 3; ++a) c[a] = sah_centroid1(d.mn[a], d.mx[a]);
        const float ca = ax.axis == 0 ? c[0] : (ax.axis == 1 ? c[1] : c[2]);
        const bool below = valid && (fallback ? p < mid : sah_bucket(ca, ax.lo, ax.extent) <= split);
        uint32_t total;
        const uint32_t bp = running + block_pred_scan256(below, wave_tot, total);
        running += total;
        if (valid) {
            uint32_t m = 0;
            if (p < mid && !below) { const uint32_t k = (p - first) - bp; tmp[first + k] = d; m = k + 1u; }
            else if (p >= mid && below) { const uint32_t k = n_below - bp - 1u; tmp[last - 1u - k] = d; m = (k + 1u) | 0x80000000u; }
            mv[p] = m;
            const int s = below ? 0 : 1;
#pragma unroll
            for (int a = 0; a < 3; ++a) {
                const uint32_t o = ordered_of(c[a]);
                if (s == 0) { lo[0][a] = min(lo[0][a], o); hi[0][a] = max(hi[0][a], o); }
                else        { lo[1][a] = min(lo[1][a], o); hi[1][a] = max(hi[1][a], o); }
            }
        }
    }
    // children's centroid bounds (only children that are split on the next level have a row)
#pragma unroll
    for (int s = 0; s < 2; ++s) {
        if (N.child[s] == kNoNode) continue;
#pragma unroll
        for (int a = 0; a < 3; ++a) {
            uint32_t l = lo[s][a], h = hi[s][a];
#pragma unroll
            for (int off = 32; off > 0; off >>= 1) {
                l = min(l, (uint32_t)__shfl_xor((int)l, off, 64));
                h = max(h, (uint32_t)__shfl_xor((int)h, off, 64));
            }
            if ((threadIdx.x & 63u) == 0u) { atomicMin(&s_cb[s][a], l); atomicMax(&s_cb[s][3 + a], h); }
        }
    }
    __syncthreads();
    if (threadIdx.x < 12u) {
        const uint32_t s = threadIdx.x / 6u, a = threadIdx.x % 6u;
        if (N.child[s] != kNoNode) {
            if (a < 3u) atomicMin(&next_nodes[N.child[s]].cb[a], s_cb[s][a]); else atomicMax(&next_nodes[N.child[s]].cb[a], s_cb[s][a]);
        }
    }
}

__global__ void __launch_bounds__(256) k_sah_apply(DLeaf* elems, const DLeaf* tmp, const uint32_t* mv, const SNode* nodes, SNode* next_nodes, const uint32_t* task_node) {
    const uint32_t t = blockIdx.x;
    const SNode& N = nodes[task_node[t]];
    if (t == N.task0 && threadIdx.x < 140u) {                              // the node's first task: empty bins for the children's rows
        const uint32_t side = threadIdx.x / 70u, w = threadIdx.x % 70u;
        if (N.child[side] != kNoNode) {
            SNode& c = next_nodes[N.child[side]];
            if (w < kSahBuckets) c.count[w] = 0u; else c.bb[(w - kSahBuckets) / 6u][(w - kSahBuckets) % 6u] = ((w - kSahBuckets) % 6u) < 3u ? 0xFFFFFFFFu : 0u;
        }
    }
    if (N.fallback) return;
    const uint32_t first = N.first, last = N.last;
    const uint32_t p0 = first + (t - N.task0) * kChunk, p1 = min(last, p0 + kChunk);
    for (uint32_t p = p0 + threadIdx.x; p < p1; p += 256u) {
        const uint32_t m = mv[p];
        if (!m) continue;
        const uint32_t k = (m & 0x7FFFFFFFu) - 1u;
        elems[p] = (m >> 31) ? tmp[first + k] : tmp[last - 1u - k];       // a position on the right takes the k-th record from the left
    }
}

__global__ void __launch_bounds__(256) k_sah_vals(const DLeaf* elems, uint32_t n, uint32_t* vals) {
    const uint32_t i = blockIdx.x * 256u + threadIdx.x;
    if (i < n) vals[i] = elems[i]._pad;
}

// One wavefront per subtree of <= 64 leaves: lane p = position p of the subtree's range.  Every level of the subtree is one
// trip of the loop, all of its nodes side by side (a lane works for the node its position is in).
__global__ void __launch_bounds__(64) k_sah_finish(const DLeaf* elems, const SFinish* finish, uint32_t* vals, uint32_t* counters, DTopo tp, uint32_t n) {
    constexpr uint32_t kWideNodes = 21;                          // nodes of >= 3 leaves among 64 positions
    __shared__ float s_c[3][64], s_mn[3][64], s_mx[3][64];      // by record (the lane that loaded it)
    __shared__ uint32_t s_ord[64], s_x[64];                      // by position: the record there; exchange
    __shared__ uint32_t s_bins[kWideNodes * 70];                 // per node of >= 3 leaves: 10 counts, 10 x 6 ordered bounds
    __shared__ float s_cost[kWideNodes * 9];
    const uint32_t lane = threadIdx.x;
    const SFinish f = finish[blockIdx.x];
    const uint32_t F = f.first, S = f.last - f.first;
    uint32_t leaf = 0;
    if (lane < S) {
        const DLeaf d = elems[F + lane];
        leaf = d._pad;
#pragma unroll
        for (int a = 0; a < 3; ++a) { s_mn[a][lane] = d.mn[a]; s_mx[a][lane] = d.mx[a]; s_c[a][lane] = sah_centroid1(d.mn[a], d.mx[a]); }
    }
    s_ord[lane] = lane;
    uint32_t nf = 0, nl = S, nq = f.qbase, depth = f.depth;
    __syncthreads();
    for (;; ++depth) {
        const uint32_t span = nl - nf;
        const bool active = lane < S && span >= 2u;
        if (__ballot(active) == 0ull) break;
        // centroid bounds of the node: a scan that stops at the node's first position, then the last position's value
        const uint32_t me = s_ord[lane < S ? lane : 0u];
        float c[3], lo[3], hi[3];
#pragma unroll
        for (int a = 0; a < 3; ++a) { c[a] = s_c[a][me]; lo[a] = c[a]; hi[a] = c[a]; }
#pragma unroll
        for (int off = 1; off < 64; off <<= 1) {
            const bool take = lane >= nf + (uint32_t)off;
#pragma unroll
            for (int a = 0; a < 3; ++a) {
                const float tl = __shfl_up(lo[a], off, 64), th = __shfl_up(hi[a], off, 64);
                if (take) { lo[a] = fminf(lo[a], tl); hi[a] = fmaxf(hi[a], th); }
            }
        }
#pragma unroll
        for (int a = 0; a < 3; ++a) { lo[a] = __shfl(lo[a], (int)((nl - 1u) & 63u), 64); hi[a] = __shfl(hi[a], (int)((nl - 1u) & 63u), 64); }
        const uint32_t axis = sah_widest(hi[0] - lo[0], hi[1] - lo[1], hi[2] - lo[2]);
        const float alo = axis == 0 ? lo[0] : (axis == 1 ? lo[1] : lo[2]);
        const float extent = (axis == 0 ? hi[0] : (axis == 1 ? hi[1] : hi[2])) - alo;
        const float my_key = axis == 0 ? c[0] : (axis == 1 ? c[1] : c[2]);
        uint32_t mid = nf + 1u;
        bool below = false, moved = false;
        uint32_t rank = 0;
        {                                                                                  // two leaves, BVH.hh:59-77
            const float ka = __shfl(my_key, (int)(nf & 63u), 64), kb = __shfl(my_key, (int)((nf + 1u) & 63u), 64);
            if (active && span == 2u && !(ka < kb)) { moved = true; rank = 0; below = lane != nf; }      // the two change places
        }
        const bool wide = active && span > 2u;
        const bool degenerate = !(extent > 0.0f);
        const bool binned = wide && !degenerate;
        // bucket rows of the nodes that are split by cost: slot = wide nodes before this one
        const unsigned long long heads = __ballot(wide && lane == nf);
        const uint32_t slot = (uint32_t)__popcll(heads & ((1ull << nf) - 1ull));
        uint32_t* bins = s_bins + slot * 70u;
        const uint32_t bk = binned ? sah_bucket(my_key, alo, extent) : 0u;
        if (binned)
            for (uint32_t w = lane - nf; w < 70u; w += span) bins[w] = (w >= kSahBuckets && ((w - kSahBuckets) % 6u) < 3u) ? 0xFFFFFFFFu : 0u;
        __syncthreads();
        if (binned) {
            atomicAdd(&bins[bk], 1u);
            uint32_t* bb = bins + kSahBuckets + bk * 6u;
#pragma unroll
            for (int a = 0; a < 3; ++a) { atomicMin(&bb[a], ordered_of(s_mn[a][me])); atomicMax(&bb[3 + a], ordered_of(s_mx[a][me])); }
        }
        __syncthreads();
        // the nine costs: the lane at offset o of a node takes candidates o, o + span, ...
        {
            const float denom = sah_area(lo, hi);
            for (uint32_t i = lane - nf; binned && i < kSahBuckets - 1u; i += span) {
                float mn0[3] = {FLT_MAX, FLT_MAX, FLT_MAX}, mx0[3] = {-FLT_MAX, -FLT_MAX, -FLT_MAX};
                float mn1[3] = {FLT_MAX, FLT_MAX, FLT_MAX}, mx1[3] = {-FLT_MAX, -FLT_MAX, -FLT_MAX};
                int c0 = 0, c1 = 0;
#pragma unroll
                for (uint32_t j = 0; j < kSahBuckets; ++j) {
                    const uint32_t cnt = bins[j];
                    if (!cnt) continue;
                    const uint32_t* bb = bins + kSahBuckets + j * 6u;
                    if (j <= i) { for (int a = 0; a < 3; ++a) { mn0[a] = fminf(mn0[a], float_of(bb[a])); mx0[a] = fmaxf(mx0[a], float_of(bb[3 + a])); } c0 += (int)cnt; }
                    else        { for (int a = 0; a < 3; ++a) { mn1[a] = fminf(mn1[a], float_of(bb[a])); mx1[a] = fmaxf(mx1[a], float_of(bb[3 + a])); } c1 += (int)cnt; }
                }
                s_cost[slot * 9u + i] = 1 + ((float)c0 * sah_area(mn0, mx0) + (float)c1 * sah_area(mn1, mx1)) / denom;
            }
        }
        __syncthreads();
        uint32_t split = 0;
        if (binned) {
            float best = 0.0f;
            for (uint32_t i = 0; i + 1 < kSahBuckets; ++i) {
                const float cost = s_cost[slot * 9u + i];
                if (i == 0 || cost < best) { best = cost; split = i; }
            }
            below = bk <= split;
        }
        const unsigned long long node_mask = (span >= 64u ? ~0ull : ((1ull << span) - 1ull)) << nf;
        const unsigned long long bm = __ballot(wide && !degenerate && below) & node_mask;
        if (wide) {
            const uint32_t n_below = (uint32_t)__popcll(bm);
            mid = nf + n_below;
            bool fallback = degenerate || mid <= nf || mid >= nl;
            if (fallback) {
                if (!degenerate && lane == nf) atomicOr(&counters[1], 16u);                // see the header: cannot happen
                mid = nf + span / 2u;
            } else {
                const unsigned long long lt = (1ull << lane) - 1ull;
                if (lane < mid && !below) { moved = true; rank = (uint32_t)__popcll(~bm & node_mask & lt); }
                else if (lane >= mid && below) { moved = true; rank = (uint32_t)__popcll(bm & ~lt & ~(1ull << lane)); }
            }
        }
        // the k-th misplaced record from the left changes places with the k-th from the right (BVH.hh:154-168)
        const bool from_right = moved && below;
        if (moved && !from_right) s_x[nf + rank] = me;
        if (from_right) s_x[nl - 1u - rank] = me;
        __syncthreads();
        if (moved) s_ord[lane] = from_right ? s_x[nf + rank] : s_x[nl - 1u - rank];
        __syncthreads();
        if (active && lane == nf) {
            const uint32_t self = sah_id_of(nq + span - 2u, n);
            const uint32_t sl = mid - nf, sr = nl - mid, rq = nq + sl - 1u;
            uint32_t l, r;
            if (sl == 1u) { l = kChildLeaf | (F + nf); tp.parent_leaf[F + nf] = self; }
            else { l = sah_id_of(nq + sl - 2u, n); tp.parent_interior[l] = self; }
            if (sr == 1u) { r = kChildLeaf | (F + mid); tp.parent_leaf[F + mid] = self; }
            else { r = sah_id_of(rq + sr - 2u, n); tp.parent_interior[r] = self; }
            tp.child_l[self] = l; tp.child_r[self] = r; tp.axis[self] = axis;
            if (self == 0u) tp.parent_interior[0] = 0u;
        }
        if (active) {
            if (lane < mid) nl = mid; else { nq = nq + (mid - nf) - 1u; nf = mid; }
        }
    }
    if (lane == 0u) atomicMax(&counters[2], depth);      // every trip put its nodes' children one level further down
    // leaf indices in their final order
    __shared__ uint32_t s_leaf[64];
    s_leaf[lane] = leaf;
    __syncthreads();
    if (lane < S) vals[F + lane] = s_leaf[s_ord[lane]];
}

// Topology (tp) and leaf order (vals[position] = leaf index) of the SAH tree over leaves[0, n).  bad = the leaf-intake flag word.
static trc_status sah_build_topology(trc_ctx* ctx, Buffers& buf, const DLeaf* d_leaves, uint32_t n, uint32_t* d_bad, DTopo tp, uint32_t* d_vals, uint32_t* out_height) {
    hipStream_t st = ctx->stream;
    const uint32_t max_rows = n / kFinishSpan + 2u, max_tasks = n / kChunk + max_rows + 2u;
    const uint32_t n_counters = 2u + 2u * (TRC_MAX_BVH_DEPTH + 3u);
    SahBufs b{};
    b.n = n;
    HIP_TRY(ctx, buf.alloc(&b.elems, n)); HIP_TRY(ctx, buf.alloc(&b.tmp, n)); HIP_TRY(ctx, buf.alloc(&b.mv, n));
    for (int k = 0; k < 2; ++k) { HIP_TRY(ctx, buf.alloc(&b.nodes[k], max_rows)); HIP_TRY(ctx, buf.alloc(&b.task_node[k], max_tasks)); }
    HIP_TRY(ctx, buf.alloc(&b.task_hist, (size_t)max_tasks * kSahBuckets));
    HIP_TRY(ctx, buf.alloc(&b.finish, n / 2u + 2u));
    HIP_TRY(ctx, buf.alloc(&b.counters, n_counters));
    HIP_TRY(ctx, hipMemsetAsync(b.counters, 0, sizeof(uint32_t) * n_counters, st));
    HIP_TRY(ctx, hipMemsetAsync(tp.parent_interior, 0, sizeof(uint32_t), st));
    const dim3 g_leaf((n + 255) / 256), b256(256);
    uint32_t n_rows = 0, n_tasks = 0;
    if (n <= kFinishSpan) {
        const SFinish f{0u, n, 0u, 0u};                 // the root, depth 0
        const uint32_t one = 1u;
        HIP_TRY(ctx, hipMemcpyAsync(b.finish, &f, sizeof f, hipMemcpyHostToDevice, st));
        HIP_TRY(ctx, hipMemcpyAsync(b.counters, &one, sizeof one, hipMemcpyHostToDevice, st));
        HIP_TRY(ctx, hipStreamSynchronize(st));          // f and one are on this frame
        hipLaunchKernelGGL(k_sah_init, g_leaf, b256, 0, st, d_leaves, n, b.elems, d_bad, (SNode*)nullptr);
    } else {
        n_rows = 1; n_tasks = (n + kChunk - 1) / kChunk;
        hipLaunchKernelGGL(k_sah_root_row, dim3(1), dim3(128), 0, st, b.nodes[0], n);
        HIP_TRY(ctx, hipMemsetAsync(b.task_node[0], 0, sizeof(uint32_t) * n_tasks, st));
        hipLaunchKernelGGL(k_sah_init, g_leaf, b256, 0, st, d_leaves, n, b.elems, d_bad, b.nodes[0]);
    }
    int cur = 0;
    bool broken = false;
    static_assert(2u + 2u * (TRC_MAX_BVH_DEPTH + 3u) + 1u <= 256u, "h_readback holds the counter block and the intake flags");
    if (!ctx->h_readback) HIP_TRY(ctx, hipHostMalloc((void**)&ctx->h_readback, 256 * sizeof(uint32_t), hipHostMallocDefault));
    uint32_t* hc = ctx->h_readback;      // pinned: the copy is queued behind k_sah_split and the host goes on launching
    hc[0] = 0;
    for (uint32_t level = 0; n_rows > 0; ++level, cur ^= 1) {
        if (level > TRC_MAX_BVH_DEPTH) return trc_fail(ctx, TRC_ERR_BVH_INVALID, "sah: tree deeper than TRC_MAX_BVH_DEPTH");
        uint32_t* next_counts = b.counters + 2u + 2u * (level + 1u);
        hipLaunchKernelGGL(k_sah_bins, dim3(n_tasks), b256, 0, st, b.elems, b.nodes[cur], b.task_node[cur], b.task_hist);
        hipLaunchKernelGGL(k_sah_split, dim3((n_rows + 63u) / 64u), dim3(64), 0, st, b.nodes[cur], n_rows, b.nodes[cur ^ 1], next_counts, b.finish, b.counters, tp, n, level);
        // one read-back per level: the whole counter block (next level's rows / tasks, error bits)
        HIP_TRY(ctx, hipMemcpyAsync(hc, b.counters, sizeof(uint32_t) * n_counters, hipMemcpyDeviceToHost, st));
        hc[n_counters] = 0;
        if (level == 0) HIP_TRY(ctx, hipMemcpyAsync(hc + n_counters, d_bad, sizeof(uint32_t), hipMemcpyDeviceToHost, st));      // intake flags are final by now
        hipLaunchKernelGGL(k_sah_scatter, dim3(n_tasks), b256, 0, st, b.elems, b.tmp, b.mv, b.nodes[cur], b.nodes[cur ^ 1], b.task_node[cur], b.task_hist, b.task_node[cur ^ 1]);
        hipLaunchKernelGGL(k_sah_apply, dim3(n_tasks), b256, 0, st, b.elems, b.tmp, b.mv, b.nodes[cur], b.nodes[cur ^ 1], b.task_node[cur]);
        HIP_TRY(ctx, hipStreamSynchronize(st));
        const uint32_t bad = hc[n_counters];
        if (bad & 8u) return trc_fail(ctx, TRC_ERR_BVH_INVALID, "sah: leaf box not finite or beyond 1e37");
        if (bad) { n_rows = 0; broken = true; break; }     // the other intake errors are reported by the caller's refit
        if (hc[1] & 16u) return trc_fail(ctx, TRC_ERR_BVH_INVALID, "sah: one-sided partition on a positive extent");
        const uint32_t next[2] = {hc[2u + 2u * (level + 1u)], hc[3u + 2u * (level + 1u)]};
        n_rows = next[0]; n_tasks = next[1];
        if (n_rows > max_rows || n_tasks > max_tasks) return trc_fail(ctx, TRC_ERR_BVH_INVALID, "sah: level table overflow");
    }
    hipLaunchKernelGGL(k_sah_vals, g_leaf, b256, 0, st, b.elems, n, d_vals);
    if (broken) return TRC_OK;
    const uint32_t n_finish = n <= kFinishSpan ? 1u : hc[0];       // every level's read-back came after its k_sah_split
    if (n_finish > n / 2u + 2u) return trc_fail(ctx, TRC_ERR_BVH_INVALID, "sah: finish list overflow");
    if (n_finish) hipLaunchKernelGGL(k_sah_finish, dim3(n_finish), dim3(64), 0, st, b.elems, b.finish, d_vals, b.counters, tp, n);
    uint32_t* flags = hc;
    HIP_TRY(ctx, hipMemcpyAsync(flags, b.counters, 3 * sizeof(uint32_t), hipMemcpyDeviceToHost, st));
    HIP_TRY(ctx, hipStreamSynchronize(st));
    *out_height = flags[2];
    if (flags[1] & 16u) return trc_fail(ctx, TRC_ERR_BVH_INVALID, "sah: one-sided partition on a positive extent");
    HIP_TRY(ctx, hipGetLastError());
    return TRC_OK;
}
