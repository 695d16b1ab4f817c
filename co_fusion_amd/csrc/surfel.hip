// surfel.hip -- the surfel half of the hot path as HIP compute kernels (no OpenGL, no interop).
//
// MI355X-native replacement of the GLSL passes driven by Core/Model/Model.cpp and
// Core/Model/ModelProjection.cpp:
//   depth_bilateral_metric.frag  -> bilateral_kernel
//   vertex_feedback.* / init     -> feedback_kernel + ordered compaction + init_kernel
//   index_map.*                  -> index_splat_kernel (64-bit atomicMin z|id keys) + index_resolve_kernel
//   splat.vert/combo_splat.frag  -> splat_raster_kernel (atomicMin) + splat_resolve_kernel
//   fill_*.frag                  -> fill_in_kernel
//   data.* + update.vert         -> associate_kernel (+ owner atomicMin) + update_kernel
//   copy_unstable.*              -> clean_kernel + ordered (stable) stream compaction
//
// The rasteriser's "nearest fragment wins, first primitive wins ties" becomes an atomicMin on a
// 64-bit key (order-preserving depth bits << 32 | surfel id); the transform-feedback append becomes
// a three-phase exclusive scan that preserves primitive order, so surfel ids and counts are exactly
// those of the reference's pipeline.  All passes are streaming / gather-scatter and HBM-bound:
// 48 B surfel records are read as 3 x float4 (16 B/lane), images are walked row-major by 256-thread
// workgroups.  Arithmetic is bit-identical to the CPU oracle (same operation order, no FMA contraction).
#include "cf_surfel_device.h"
#include "cf_kernels.h"

namespace cf {

static constexpr int kB = 256;
static inline int gridFor(long long n) { return (int)((n + kB - 1) / kB); }
// XCD-aware workgroup order for passes that gather from the index-map images: the hardware deals consecutive workgroups round-robin
// over the 8 XCDs, each with its own L2, so neighbouring workgroups -- neighbouring surfels / pixels, which read the same image
// lines -- would pull every line into all eight L2s.  With a grid that is a multiple of 8, XCD x is given the x-th contiguous range
// of logical workgroups (a band of the image / a run of the surfel buffer) instead.
static inline int gridXcd(long long n, int chunk = 1) { const int q = 8 * (chunk > 1 ? chunk : 1); return (gridFor(n) + q - 1) / q * q; }  // xcd_block is a bijection on it
// chunk > 0: the XCDs take turns on runs of `chunk` consecutive logical workgroups (locality inside a run, balance between the XCDs
// when the cost per workgroup drifts along the buffer); chunk < 0: one band per XCD; 0: plain round-robin (the hardware's order).
// (the orders in use were chosen by measurement, DESIGN-NOTES.md; a diagnostics build -- make ABLATE=1 -- reads them from the environment)
#ifdef CF_ABLATE
static int xcd_env(const char* name, int dflt) { const char* e = getenv(name); return e ? atoi(e) : dflt; }
#else
static constexpr int xcd_env(const char*, int dflt) { return dflt; }
#endif
// LOCK-STEP BATCHES (round 4).  The surfel passes of a frame are the same chain of ~16 kernels for every model, and an object model's
// share of each is a few dozen workgroups: as one chain per model on its own stream (rounds 1-3) the five chains of configs[2] took
// 450 us of a 1.5 ms frame -- the HIP streams share a handful of hardware queues, so chains ran one after another, 100 launch-floor
// kernels in all (profiles/r4e timeline).  Now every stage is ONE launch for all models of the batch, exactly as the tracking
// launches have been since round 1: a one-dimensional grid [model 0's workgroups | model 1's | ...], the running totals in the
// kernel arguments (BatchHdr), each model's arguments beside them.  A workgroup finds its model with 15 scalar compares and then
// runs the unchanged per-model code on its virtual block index.  Every model's share starts at a multiple of 8 workgroups, so the
// XCD of a workgroup (hardware id mod 8) is also that of its virtual index (xcd_block).
struct BatchHdr { int n; int blk_end[kSurfBatch]; };
template <class A> struct Batch { BatchHdr h; A m[kSurfBatch]; };
struct VBlock { int model, bid, nblk; };
__device__ __forceinline__ VBlock batch_decode_at(const BatchHdr& h, int b)
{
    int slot = 0;
#pragma unroll
    for (int k = 0; k < kSurfBatch - 1; k++) slot += (b >= h.blk_end[k]) ? 1 : 0;
    const int start = slot ? h.blk_end[slot - 1] : 0;
    return VBlock{slot, b - start, h.blk_end[slot] - start};
}
__device__ __forceinline__ VBlock batch_decode(const BatchHdr& h) { return batch_decode_at(h, (int)blockIdx.x); }
// host side: the table of a launch from the models' workgroup counts (each padded to a multiple of 8); returns the grid size
template <class A>
static int batch_layout(Batch<A>& B, const int* blocks, int n)
{
    int total = 0;
    B.h.n = n;
    for (int k = 0; k < kSurfBatch; k++) {
        if (k < n) total += (blocks[k] + 7) / 8 * 8;
        B.h.blk_end[k] = k < n ? total : 0x7fffffff;
    }
    return total;
}

__device__ __forceinline__ int xcd_block(int chunk, int b, int nblk)
{
    if (chunk == 0) return b;
    const int per = nblk >> 3, x = b & 7, r = b >> 3;  // XCD, rank inside the XCD
    if (chunk < 0) return x * per + r;
    const int lb = ((r / chunk) * 8 + x) * chunk + (r % chunk);
    return lb;  // (the grid is a multiple of 8 * chunk: a bijection)
}

__device__ __forceinline__ unsigned sortable_bits(float z)
{  // order-preserving float -> uint
    const unsigned b = __float_as_uint(z);
    return (b & 0x80000000u) ? ~b : (b | 0x80000000u);
}
__device__ __forceinline__ unsigned long long zkey(float z, unsigned id) { return ((unsigned long long)sortable_bits(z) << 32) | id; }
constexpr unsigned long long kEmptyKey = ~0ull;

// ================================================================================ bilateral ====
// depth_bilateral_metric.frag:30-75
__global__ void __launch_bounds__(kB) bilateral_kernel(const float* __restrict__ depth, int cols, int rows, float maxD,
                                                       float* __restrict__ out)
{
    const int i = blockIdx.x * kB + threadIdx.x;
    if (i >= cols * rows) return;
    const int y = i / cols, x = i - y * cols;
    const float value = depth[i];
    if (value > maxD || value < 0.3f) { out[i] = 0; return; }
    // The kernel is VALU-bound (169 taps x exp per pixel), so the 13x13 window is fully unrolled: the spatial term
    // ((float)x-(float)cx)^2 + ((float)y-(float)cy)^2 is an exact small integer, hence space2 * 0.024691358f folds to
    // a per-tap constant with the same value the shader computes at run time.  Taps outside the image are skipped
    // (the shader's loop bounds), in the same row-major order.
    float sum1 = 0, sum2 = 0;
#pragma unroll
    for (int dy = -6; dy <= 6; ++dy) {
        const int cy = y + dy;
        if (cy < 0 || cy >= rows) continue;
        const float* __restrict__ rowp = depth + cy * cols + x;
#pragma unroll
        for (int dx = -6; dx <= 6; ++dx) {
            const int cx = x + dx;
            if (cx < 0 || cx >= cols) continue;
            const float tmp = rowp[dx];
            const float space2 = (float)(dx * dx + dy * dy);
            const float color2 = (value - tmp) * (value - tmp);
            const float weight = det_expf(-(space2 * 0.024691358f + color2 * 555.556f));
            sum1 += tmp * weight;
            sum2 += weight;
        }
    }
    out[i] = sum1 / sum2;
}

// ==================================================================== ordered compaction (scan) ====
// Ordered compaction of flagged records in blocks of kScanItems elements: block sums, then scan-and-scatter.
static constexpr int kScanItems = 2048;  // 256 threads x 8
struct ScanArgs {   // one ordered compaction: flags [n] -> block sums -> the flagged 48 B records of rec moved to out, in order
    const float4* rec; const unsigned* flags; long long n; unsigned* block_sums; unsigned* total; unsigned add_to_total; float4* out;
    unsigned* total_host; int zero_flags;
};
// eight consecutive flags of a thread (i0 = element index of the first); one flight of loads: two 16-byte loads for a full group, the
// guarded tail otherwise
__device__ __forceinline__ void scan_load_flags(const unsigned* __restrict__ flags, long long i0, long long n, unsigned (&f)[8])
{
    if (i0 + 8 <= n) {
        const uint4 a = *reinterpret_cast<const uint4*>(flags + i0), c = *reinterpret_cast<const uint4*>(flags + i0 + 4);
        f[0] = a.x; f[1] = a.y; f[2] = a.z; f[3] = a.w; f[4] = c.x; f[5] = c.y; f[6] = c.z; f[7] = c.w;
    } else {
        unsigned t[8];
#pragma unroll
        for (int e = 0; e < 8; e++) t[e] = flags[i0 + e < n ? i0 + e : n - 1];   // (unconditional, address clamped: a load under a branch is waited for at its end)
#pragma unroll
        for (int e = 0; e < 8; e++) f[e] = (i0 + e < n) ? t[e] : 0u;
    }
}
__device__ __forceinline__ void scan_block_sums_body(const Batch<ScanArgs>& B, int blk)
{
    const VBlock vb = batch_decode_at(B.h, blk);
    const ScanArgs& a = B.m[vb.model];
    const unsigned* __restrict__ flags = a.flags; const long long n = a.n; unsigned* __restrict__ block_sums = a.block_sums;
    if ((long long)vb.bid * kScanItems >= n) return;   // (padding workgroups of the batch layout)
    // (round 6: the thread's eight flags as two vector loads -- eight guarded loads had been eight dependent round trips in the binary)
    unsigned f[8];
    scan_load_flags(flags, (long long)vb.bid * kScanItems + (long long)threadIdx.x * 8, n, f);
    unsigned s = 0;
#pragma unroll
    for (int e = 0; e < 8; e++) s += f[e];
    for (int o = 32; o > 0; o >>= 1) s += __shfl_xor((int)s, o, 64);
    __shared__ unsigned w[4];
    if ((threadIdx.x & 63) == 0) w[threadIdx.x >> 6] = s;
    __syncthreads();
    if (threadIdx.x == 0) block_sums[vb.bid] = w[0] + w[1] + w[2] + w[3];
}
__global__ void __launch_bounds__(256) scan_block_sums_kernel(const Batch<ScanArgs> B) { scan_block_sums_body(B, (int)blockIdx.x); }
__global__ void set_count_kernel(unsigned* out, unsigned v);
__global__ void set_count2_kernel(unsigned* out, unsigned* out_host, unsigned v);
// Ordered compaction in TWO launches (scan_block_sums_kernel, then this): every workgroup derives its own base from the block
// sums of the workgroups before it (at most a few hundred values), scans its 2048 flags -- eight consecutive flags per thread, so
// one pass and two barriers -- and moves the flagged 48 B records to their place; the last workgroup publishes the total.
// Replaces {spine scan, per-element offsets, scatter} = three launches and an offsets array.
// Round 6, the move: until then every thread copied its own flagged records one after the other -- up to eight dependent load -> store
// round trips, reads 384 bytes apart between neighbouring lanes.  Now the threads list the block's flagged elements in LDS (the
// compaction keeps the order, so they fill ONE contiguous range of the output) and the workgroup copies that range as a stream of
// 16-byte words: thread t writes word t, t + 256, ... of the range -- contiguous stores, reads contiguous wherever the flags are dense --
// twelve independent loads in flight per thread.
__device__ __forceinline__ void scan_scatter_body(const Batch<ScanArgs>& B, int blk)
{
    const VBlock vb = batch_decode_at(B.h, blk);
    const ScanArgs& a = B.m[vb.model];
    const float4* __restrict__ rec = a.rec; const unsigned* __restrict__ flags = a.flags; const long long n = a.n;
    const unsigned* __restrict__ block_sums = a.block_sums; unsigned* __restrict__ total = a.total; const unsigned add_to_total = a.add_to_total;
    float4* __restrict__ out = a.out; unsigned* __restrict__ total_host = a.total_host;
    const int nb = (int)((n + kScanItems - 1) / kScanItems);   // workgroups that hold elements (the batch layout pads to a multiple of 8)
    if (vb.bid >= nb) return;
    __shared__ unsigned wsum[4], s_base;
    __shared__ unsigned short s_src[kScanItems];   // element (within the block) behind every output slot of the block
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const long long i0 = (long long)vb.bid * kScanItems + (long long)tid * 8;
    unsigned f[8];
    scan_load_flags(flags, i0, n, f);   // (in the same flight as the block sums below)
    if (a.zero_flags) {   // the flags are consumed: leave zeros behind for the next pass that sets some of them (no fill launch per frame)
        unsigned* fz = const_cast<unsigned*>(flags);
        if (i0 + 8 <= n) { *reinterpret_cast<uint4*>(fz + i0) = make_uint4(0, 0, 0, 0); *reinterpret_cast<uint4*>(fz + i0 + 4) = make_uint4(0, 0, 0, 0); }
        else
            for (int e = 0; e < 8; e++) if (i0 + e < n) fz[i0 + e] = 0;
    }
    unsigned b = 0;
    for (int k = tid; k < vb.bid; k += 256) b += block_sums[k];
    for (int o = 32; o > 0; o >>= 1) b += __shfl_xor((int)b, o, 64);
    if (lane == 0) wsum[wave] = b;
    __syncthreads();
    if (tid == 0) s_base = wsum[0] + wsum[1] + wsum[2] + wsum[3];
    unsigned mine = 0;
#pragma unroll
    for (int e = 0; e < 8; e++) mine += f[e];   // (flags are 0 / 1)
    unsigned incl = mine;
    for (int o = 1; o < 64; o <<= 1) { const unsigned t = __shfl_up((int)incl, o, 64); if (lane >= o) incl += t; }
    __syncthreads();  // s_base written, wsum free again
    if (lane == 63) wsum[wave] = incl;
    __syncthreads();
    unsigned rel = incl - mine;
    for (int w = 0; w < wave; w++) rel += wsum[w];
#pragma unroll
    for (int e = 0; e < 8; e++)
        if (f[e]) s_src[rel++] = (unsigned short)(tid * 8 + e);
    const unsigned count = wsum[0] + wsum[1] + wsum[2] + wsum[3];
    const unsigned base = s_base;
    __syncthreads();
    const float4* __restrict__ in0 = rec + (size_t)vb.bid * kScanItems * 3;
    float4* __restrict__ out0 = out + (size_t)base * 3;
    const unsigned words = count * 3;
#ifndef CF_SCAN_FLY
#define CF_SCAN_FLY 4
#endif
    constexpr int kFly = CF_SCAN_FLY;   // 16-byte words in flight per thread (a full block is 24 per thread)
    for (unsigned m0 = tid; m0 < words; m0 += kFly * 256) {
        float4 v[kFly];
#pragma unroll
        for (int u = 0; u < kFly; u++) {
            const unsigned m = m0 + u * 256;
            const unsigned mc = m < words ? m : words - 1;      // (clamped: the loads stay one flight)
            const unsigned j = mc / 3u;
            v[u] = in0[(size_t)s_src[j] * 3 + (mc - 3u * j)];
        }
#pragma unroll
        for (int u = 0; u < kFly; u++) { const unsigned m = m0 + u * 256; if (m < words) out0[m] = v[u]; }
    }
    if (vb.bid == nb - 1 && tid == 255) {  // the last workgroup's end is the total
        *total = base + count + add_to_total;
        if (total_host) *total_host = base + count + add_to_total;  // pinned host memory: the read-back needs no copy command on the stream
    }
}
__global__ void __launch_bounds__(256) scan_scatter_kernel(const Batch<ScanArgs> B) { scan_scatter_body(B, (int)blockIdx.x); }

// n models' compactions in two launches; a model without elements (n == 0) only has its total set
void launch_scan_scatter_batch(hipStream_t s, const ScanPassArgs* items, int n_items)
{
    for (int base = 0; base < n_items; base += kSurfBatch) {
        const int nb = n_items - base < kSurfBatch ? n_items - base : kSurfBatch;
        Batch<ScanArgs> B;
        int blocks[kSurfBatch], any = 0;
        for (int k = 0; k < nb; k++) {
            const ScanPassArgs& h = items[base + k];
            B.m[k] = ScanArgs{reinterpret_cast<const float4*>(h.rec), h.flags, h.n, h.block_sums, h.total, h.add_to_total, reinterpret_cast<float4*>(h.out), h.total_host, h.zero_flags};
            blocks[k] = (int)((B.m[k].n + kScanItems - 1) / kScanItems);
            any += blocks[k];
            if (blocks[k] == 0) set_count2_kernel<<<1, 1, 0, s>>>(B.m[k].total, B.m[k].total_host, B.m[k].add_to_total);
        }
        if (!any) continue;
        const int grid = batch_layout(B, blocks, nb);
        scan_block_sums_kernel<<<grid, 256, 0, s>>>(B);
        scan_scatter_kernel<<<grid, 256, 0, s>>>(B);
    }
}
void launch_scan_scatter(hipStream_t s, const float* rec, const unsigned* flags, long long n, unsigned* block_sums, unsigned* total,
                         unsigned add_to_total, float* out, unsigned* total_host)
{
    const ScanPassArgs a{rec, flags, n, block_sums, total, add_to_total, out, total_host};
    launch_scan_scatter_batch(s, &a, 1);
}

// ============================================================================ frame-1 bootstrap ====
// vertex_feedback.vert:40-68 (+ .geom): thread per pixel (row-major for coalescing); the record and
// its flag are stored at the COLUMN-major rank i*rows+j, the order the reference draws the points in.
__global__ void __launch_bounds__(kB) feedback_kernel(const uchar4* __restrict__ rgba, const float* __restrict__ depth, int cols, int rows,
                                                      float cx, float cy, float inv_fx, float inv_fy, const float* __restrict__ tcx,
                                                      const float* __restrict__ tcy, int time, float maxDepth,
                                                      float4* __restrict__ rec /* [N*3] by rank */, unsigned* __restrict__ flags)
{
    const int q = blockIdx.x * kB + threadIdx.x;
    if (q >= cols * rows) return;
    const int j = q / cols, i = q - j * cols;
    const int rank = i * rows + j;
    const float x = tcx[i] * (float)cols, y = tcy[j] * (float)rows;
    const f3 p = get_vertex(depth, cols, rows, i, j, x, y, cx, cy, inv_fx, inv_fy);
    if (p.z <= 0 || p.z > maxDepth) { flags[rank] = 0; return; }
    const f3 n = get_normal_central(p, depth, cols, rows, i, j, x, y, cx, cy, inv_fx, inv_fy);
    const uchar4 c = rgba[q];
    flags[rank] = 1;
    rec[rank * 3 + 0] = make_float4(p.x, p.y, p.z, confidence(x, y, cx, cy, 1.0f));
    rec[rank * 3 + 1] = make_float4(encode_color((float)c.x / 255.0f, (float)c.y / 255.0f, (float)c.z / 255.0f), 0.f, (float)c.z / 255.0f, (float)time);
    rec[rank * 3 + 2] = make_float4(n.x, n.y, n.z, get_radius(p.z, n.z, inv_fx, inv_fy));
}


// Model::initialise (Model.cpp:227-272) + init_unstable.vert: attr 0/1 raw, attr 2 filtered, same index
__global__ void __launch_bounds__(kB) init_kernel(const float4* __restrict__ raw, const float4* __restrict__ filt,
                                                  const unsigned* __restrict__ raw_count, float4* __restrict__ out)
{
    const unsigned k = blockIdx.x * kB + threadIdx.x;
    if (k >= *raw_count) return;
    const float4 c = raw[k * 3 + 1];
    out[k * 3] = raw[k * 3];
    out[k * 3 + 1] = make_float4(c.x, 0.f, 1.f, c.w);
    out[k * 3 + 2] = filt[k * 3 + 2];
}

// ================================================================================= index map ====
// index_map.vert:38-63
__device__ __forceinline__ bool index_project(const float4 pc, const float4 ct, const Mat4& t_inv, cf_cam cam, int cols, int rows,
                                              float maxDepth, int time, int timeDelta, f3& ph, int& q)
{
    ph = xform_point(t_inv, f3{pc.x, pc.y, pc.z});
    if (ph.z > maxDepth || ph.z < 0 || (float)time - ct.w > (float)timeDelta) return false;
    const float u = ((cam.fx * ph.x) / ph.z) + cam.cx, v = ((cam.fy * ph.y) / ph.z) + cam.cy;
    if (!(u >= 0.0f && v >= 0.0f && u < (float)cols && v < (float)rows)) return false;
    if (!(ph.z < maxDepth)) return false;  // depth buffer cleared to 1.0, GL_LESS
    q = (int)floorf(v) * cols + (int)floorf(u);
    return true;
}

struct IndexArgs {   // predictIndices of one model (both kernels)
    const float4* surfels; const unsigned* count; Mat4 t_inv; float maxDepth; int time, timeDelta; unsigned id_begin, id_end;
    unsigned long long* keys; unsigned* index; float4* vertConf; float4* colorTime; float4* normRad;
    const float* t_inv_dev;   // nullable: the matrix in device memory instead (IndexPassArgs::t_inv_dev)
    float4* clean_rec; const float* clean_depth;   // nullable: the clean pass's packed per-texel records (IndexPassArgs::clean_rec)
};
__device__ __forceinline__ Mat4 index_matrix(const IndexArgs& a)
{
    if (!a.t_inv_dev) return a.t_inv;
    Mat4 m;
#pragma unroll
    for (int k = 0; k < 16; k++) m.m[k] = a.t_inv_dev[k];   // (uniform address: one request per wave)
    return m;
}
// (the inverse of a tracked pose is left in the tracker's state by the last solve of its schedule -- OdomDev::pose_inv -- since round 6: a
// kernel of its own, pose_tinv_kernel, until then)
struct FrameGeom { cf_cam cam; int cols, rows; };   // what the models of a launch share

__device__ __forceinline__ void index_splat_body(const Batch<IndexArgs>& B, const FrameGeom& g, int blk)
{
    const VBlock vb = batch_decode_at(B.h, blk);
    const IndexArgs& a = B.m[vb.model];
    const float4* __restrict__ surfels = a.surfels;
    // [id_begin, id_end): the surfel range of this launch (the whole map, or a rank's shard of it)
    const unsigned id = a.id_begin + vb.bid * kB + threadIdx.x;
    if (id >= *a.count || id >= a.id_end) return;
    f3 ph; int q;
    const Mat4 T = index_matrix(a);
    if (!index_project(surfels[id * 3], surfels[id * 3 + 1], T, g.cam, g.cols, g.rows, a.maxDepth, a.time, a.timeDelta, ph, q)) return;
    atomicMin(&a.keys[q], zkey(ph.z, id));
}
__global__ void __launch_bounds__(kB) index_splat_kernel(const Batch<IndexArgs> B, const FrameGeom g) { index_splat_body(B, g, (int)blockIdx.x); }

__global__ void __launch_bounds__(kB) index_resolve_kernel(const Batch<IndexArgs> B, const FrameGeom g)
{
    const VBlock vb = batch_decode(B.h);
    const IndexArgs& a = B.m[vb.model];
    const float4* __restrict__ surfels = a.surfels; unsigned long long* __restrict__ keys = a.keys; unsigned* __restrict__ index = a.index;
    float4* __restrict__ vertConf = a.vertConf; float4* __restrict__ colorTime = a.colorTime; float4* __restrict__ normRad = a.normRad;
    const int q = vb.bid * kB + threadIdx.x;
    if (q >= g.cols * g.rows) return;
    const unsigned long long k = keys[q];
    keys[q] = kEmptyKey;  // leave the z-buffer cleared for the next pass (no memset launch per projection)
    // The pass in front of the clean stage also leaves what clean_kernel stages per texel as ONE 32-byte record (vertConf | colorTime.zw, index,
    // the frame's filtered depth) beside the arrays the other passes read (late in round 6: +7 us here, -17 us there).  Measured as well:
    // the two halves changing lanes so that every store instruction of a wave writes 1 KB in one piece (no gain: 23.4 against 22.8 us -- it
    // is the bytes, not the store pattern), and a 16-byte record without vertConf (clean_kernel's comment).
    float4* __restrict__ const rec = a.clean_rec;
    const float dflt = rec ? a.clean_depth[q] : 0.f;
    if (k == kEmptyKey) {
        index[q] = 0;
        vertConf[q] = colorTime[q] = normRad[q] = make_float4(0, 0, 0, 0);
        if (rec) { rec[2 * q] = make_float4(0, 0, 0, 0); rec[2 * q + 1] = make_float4(0, 0, __uint_as_float(0u), dflt); }
        return;
    }
    const unsigned id = (unsigned)k;
    const float4 pc = surfels[id * 3], ct = surfels[id * 3 + 1], nr = surfels[id * 3 + 2];
    const Mat4 T = index_matrix(a);
    const f3 ph = xform_point(T, f3{pc.x, pc.y, pc.z});
    const f3 n = normalized(xform_dir(T, f3{nr.x, nr.y, nr.z}));
    index[q] = id;
    vertConf[q] = make_float4(ph.x, ph.y, ph.z, pc.w);
    colorTime[q] = ct;
    normRad[q] = make_float4(n.x, n.y, n.z, nr.w);
    if (rec) { rec[2 * q] = make_float4(ph.x, ph.y, ph.z, pc.w); rec[2 * q + 1] = make_float4(ct.z, ct.w, __uint_as_float(id), dflt); }
}

// ============================================================================ splat prediction ====
// splat.vert:54-88 + combo_splat.frag:37-65
struct SplatSetup { f3 ph, n; float rad, pn; int x_lo, x_hi, y_lo, y_hi; };

__device__ __forceinline__ bool splat_setup(const float4 pc, const float4 ct, const float4 nr, const Mat4& t_inv, cf_cam cam, int cols,
                                            int rows, float maxDepth, float confThreshold, int time, int maxTime, int timeDelta,
                                            SplatSetup& s)
{
    s.ph = xform_point(t_inv, f3{pc.x, pc.y, pc.z});
    if (s.ph.z > maxDepth || s.ph.z < 0 || pc.w < confThreshold || (float)time - ct.w > (float)timeDelta || ct.w > (float)maxTime) return false;
    s.n = normalized(xform_dir(t_inv, f3{nr.x, nr.y, nr.z}));
    s.rad = nr.w;
    const f3 x1n = normalized(f3{s.n.y - s.n.z, -s.n.x, s.n.x});
    const f3 x1 = {x1n.x * s.rad * 1.41421356f, x1n.y * s.rad * 1.41421356f, x1n.z * s.rad * 1.41421356f};
    const f3 y1 = cross(s.n, x1);
    const f3 c[4] = {s.ph + x1, s.ph + y1, s.ph - y1, s.ph - x1};
    float px_[4], py_[4];
#pragma unroll
    for (int k = 0; k < 4; k++) { px_[k] = ((cam.fx * c[k].x) / c[k].z) + cam.cx; py_[k] = ((cam.fy * c[k].y) / c[k].z) + cam.cy; }
    const float xmin = fminf(px_[0], fminf(px_[1], fminf(px_[2], px_[3]))), xmax = fmaxf(px_[0], fmaxf(px_[1], fmaxf(px_[2], px_[3])));
    const float ymin = fminf(py_[0], fminf(py_[1], fminf(py_[2], py_[3]))), ymax = fmaxf(py_[0], fmaxf(py_[1], fmaxf(py_[2], py_[3])));
    const float size = fmaxf(0.0f, fmaxf(fabsf(xmax - xmin), fabsf(ymax - ymin)));
    if (!(size > 0.0f) || !(size <= 4096.0f)) return false;
    const float u = ((cam.fx * s.ph.x) / s.ph.z) + cam.cx, v = ((cam.fy * s.ph.y) / s.ph.z) + cam.cy;
    if (!(u >= 0.0f && v >= 0.0f && u <= (float)cols && v <= (float)rows)) return false;  // points are clipped by their centre
    const float half = size * 0.5f;
    s.x_lo = max((int)ceilf(u - half - 0.5f), 0); s.x_hi = min((int)ceilf(u + half - 0.5f) - 1, cols - 1);
    s.y_lo = max((int)ceilf(v - half - 0.5f), 0); s.y_hi = min((int)ceilf(v + half - 0.5f) - 1, rows - 1);
    s.pn = dot(s.ph, s.n);
    return true;
}

// fragment: returns false when discarded; z = corrected depth
// `rays` holds normalized((px + 0.5 - cx) / fx, (py + 0.5 - cy) / fy, 1) per pixel (splat_rays_kernel): the
// fragment shader's view ray depends only on the pixel, and its two divisions + normalisation (sqrt + 3 IEEE
// divisions) were half of the VALU work of every fragment.
__device__ __forceinline__ bool splat_fragment(const SplatSetup& s, const float4* __restrict__ rays, int cols, int px, int py, float maxDepth,
                                               float& z)
{
    const float4 lr = rays[py * cols + px];
    const f3 l = {lr.x, lr.y, lr.z};
    const float k = s.pn / dot(l, s.n);
    const f3 cp = {k * l.x, k * l.y, k * l.z};
    const f3 diff = cp - s.ph;
    if (!(dot(diff, diff) <= s.rad * s.rad)) return false;
    if (!(cp.z < maxDepth)) return false;
    z = cp.z;
    return true;
}

__global__ void __launch_bounds__(kB) splat_rays_kernel(cf_cam cam, int cols, int rows, float4* __restrict__ rays)
{
    const int q = blockIdx.x * kB + threadIdx.x;
    if (q >= cols * rows) return;
    const int py = q / cols, px = q - py * cols;
    const float fx_ = (float)px + 0.5f, fy_ = (float)py + 0.5f;
    const f3 l = normalized(f3{(fx_ - cam.cx) / cam.fx, (fy_ - cam.cy) / cam.fy, 1.0f});
    rays[q] = make_float4(l.x, l.y, l.z, 0.f);
}

struct SplatArgs {   // combinedPredict of one model (both kernels)
    const float4* surfels; const unsigned* count; Mat4 t_inv; float maxDepth, confThreshold; int time, maxTime, timeDelta;
    const float4* rays; unsigned long long* keys; uchar4* image; float4* vertexConf; float4* normalRad; unsigned short* time16;
};

__global__ void __launch_bounds__(kB) splat_raster_kernel(const Batch<SplatArgs> B, const FrameGeom g)
{
    const VBlock vb = batch_decode(B.h);
    const SplatArgs& a = B.m[vb.model];
    const float4* __restrict__ surfels = a.surfels; const float4* __restrict__ rays = a.rays; unsigned long long* __restrict__ keys = a.keys;
    const int cols = g.cols, rows = g.rows;
    // four lanes per surfel: the fragments of a point sprite are independent (the z-test is an atomicMin), and a
    // lane walking a 5x5 footprint alone is a chain of 25 dependent ray loads; the set-up is recomputed per lane
    const unsigned gt = vb.bid * kB + threadIdx.x;
    const unsigned id = gt >> 2;
    const int sub = (int)(gt & 3u);
    if (id >= *a.count) return;
    SplatSetup s;
    if (!splat_setup(surfels[id * 3], surfels[id * 3 + 1], surfels[id * 3 + 2], a.t_inv, g.cam, cols, rows, a.maxDepth, a.confThreshold, a.time,
                     a.maxTime, a.timeDelta, s))
        return;
    const int w = s.x_hi - s.x_lo + 1, h = s.y_hi - s.y_lo + 1;
    if (w <= 0 || h <= 0) return;
    int fx = sub, fy = 0;
    while (fx >= w) { fx -= w; fy++; }
    while (fy < h) {
        const int px = s.x_lo + fx, py = s.y_lo + fy;
        float z;
        if (splat_fragment(s, rays, cols, px, py, a.maxDepth, z)) {
            // (Measured and dropped, late in round 6: `if (zk < keys[q])` in front of the atomic -- a key only ever decreases, so a fragment
            // that does not beat what the pixel holds now never will, and ~11 of a pixel's ~14 atomics would go: 48.4 against 35.0 us.  The
            // fire-and-forget atomics are not what the kernel waits for; a load in every fragment's chain is.)
            atomicMin(&keys[py * cols + px], zkey(z, id));
        }
        fx += 4;
        while (fx >= w) { fx -= w; fy++; }
    }
}

__global__ void __launch_bounds__(kB) splat_resolve_kernel(const Batch<SplatArgs> B, const FrameGeom g)
{
    const VBlock vb = batch_decode(B.h);
    const SplatArgs& a = B.m[vb.model];
    const float4* __restrict__ surfels = a.surfels; const float4* __restrict__ rays = a.rays; unsigned long long* __restrict__ keys = a.keys;
    uchar4* __restrict__ image = a.image; float4* __restrict__ vertexConf = a.vertexConf; float4* __restrict__ normalRad = a.normalRad;
    unsigned short* __restrict__ time16 = a.time16;
    const int cols = g.cols, rows = g.rows;
    const cf_cam cam = g.cam;
    const int q = vb.bid * kB + threadIdx.x;
    if (q >= cols * rows) return;
    const unsigned long long k = keys[q];
    keys[q] = kEmptyKey;  // leave the z-buffer cleared for the next pass (no memset launch per projection)
    if (k == kEmptyKey) {
        image[q] = make_uchar4(0, 0, 0, 0);
        vertexConf[q] = normalRad[q] = make_float4(0, 0, 0, 0);
        time16[q] = 0;
        return;
    }
    const unsigned id = (unsigned)k;
    const int py = q / cols, px = q - py * cols;
    const float4 pc = surfels[id * 3], ct = surfels[id * 3 + 1], nr = surfels[id * 3 + 2];
    SplatSetup s;
    splat_setup(pc, ct, nr, a.t_inv, cam, cols, rows, a.maxDepth, a.confThreshold, a.time, a.maxTime, a.timeDelta, s);
    float z;
    splat_fragment(s, rays, cols, px, py, a.maxDepth, z);
    const f3 col = decode_color(ct.x);
    image[q] = make_uchar4((unsigned char)glsl_round(col.x * 255.0f), (unsigned char)glsl_round(col.y * 255.0f),
                           (unsigned char)glsl_round(col.z * 255.0f), 255);
    const float fx_ = (float)px + 0.5f, fy_ = (float)py + 0.5f;
    vertexConf[q] = make_float4((fx_ - cam.cx) * z * (1.f / cam.fx), (fy_ - cam.cy) * z * (1.f / cam.fy), z, pc.w);
    normalRad[q] = make_float4(s.n.x, s.n.y, s.n.z, s.rad);
    time16[q] = (unsigned short)(unsigned)ct.z;
}

// ==================================================================================== fill-in ====
// fill_vertex.frag / fill_normal.frag / fill_rgb.frag
__global__ void __launch_bounds__(kB) fill_in_kernel(const float4* __restrict__ pv, const float4* __restrict__ pn, const uchar4* __restrict__ pimg,
                                                     const float* __restrict__ depth, const uchar4* __restrict__ rgba, int cols, int rows,
                                                     float cx, float cy, float inv_fx, float inv_fy, int pass_geom, int pass_rgb,
                                                     float4* __restrict__ ov, float4* __restrict__ on, uchar4* __restrict__ oi)
{
    const int q = blockIdx.x * kB + threadIdx.x;
    if (q >= cols * rows) return;
    const int y = q / cols, x = q - y * cols;
    const float z = depth[q];
    const f3 p = {((float)x - cx) * z * inv_fx, ((float)y - cy) * z * inv_fy, z};
    const float4 v = pv[q];
    ov[q] = (v.z == 0 || pass_geom) ? make_float4(p.x, p.y, p.z, 1.f) : v;
    const float4 n = pn[q];
    if (n.z == 0 || pass_geom) {
        const float zx = depth[y * cols + min(x + 1, cols - 1)], zy = depth[min(y + 1, rows - 1) * cols + x];
        const f3 vx = {((float)(x + 1) - cx) * zx * inv_fx, ((float)y - cy) * zx * inv_fy, zx};
        const f3 vy = {((float)x - cx) * zy * inv_fx, ((float)(y + 1) - cy) * zy * inv_fy, zy};
        const f3 nn = normalized(cross(vx - p, vy - p));
        on[q] = make_float4(nn.x, nn.y, nn.z, 1.f);
    } else
        on[q] = n;
    const uchar4 e = pimg[q];
    oi[q] = (((int)e.x + (int)e.y + (int)e.z) == 0 || pass_rgb) ? rgba[q] : e;
}

// CoFusion::requiresFillIn (CoFusion.cpp:547-565): one workgroup counts the 20x down-sampled image
__global__ void __launch_bounds__(1024) fill_ratio_kernel(const uchar4* __restrict__ pimg, int cols, int rows, unsigned* __restrict__ out2,
                                                          unsigned* __restrict__ out2_host)
{
    const int dw = cols / 20, dh = rows / 20;
    unsigned s = 0;
    for (int k = threadIdx.x; k < dw * dh; k += 1024) {
        const int j = k / dw, i = k - j * dw;
        const int sx = nearest_texel(((float)i + 0.5f) / (float)dw, cols), sy = nearest_texel(((float)j + 0.5f) / (float)dh, rows);
        const uchar4 p = pimg[sy * cols + sx];
        s += (p.x > 0 && p.y > 0 && p.z > 0) ? 1u : 0u;
    }
    for (int o = 32; o > 0; o >>= 1) s += __shfl_xor((int)s, o, 64);
    __shared__ unsigned w[16];
    if ((threadIdx.x & 63) == 0) w[threadIdx.x >> 6] = s;
    __syncthreads();
    if (threadIdx.x == 0) {
        unsigned t = 0;
        for (int k = 0; k < 16; k++) t += w[k];
        out2[0] = t; out2[1] = (unsigned)(dw * dh);
        if (out2_host) { out2_host[0] = t; out2_host[1] = (unsigned)(dw * dh); }
    }
}

// ===================================================================================== fusion ====
// data.vert:78-211.  Thread per pixel (row-major); merge candidates race for their surfel with an
// atomicMin on the column-major rank (= "first drawn fragment wins" on the reference's update maps).
__device__ __forceinline__ float angle_between(f3 a, f3 b) { return det_acosf(dot(a, b) / (norm(a) * norm(b))); }

struct FuseArgs {
    const unsigned* index; const float4* vertConf; const float4* normRad;
    const uchar4* rgba; const float* depth_raw; const float* depth_filt; const unsigned char* mask;
    const float* tcx; const float* tcy;
    Mat4 pose; cf_cam cam; float inv_fx, inv_fy;
    int cols, rows, time; float weighting; int maskID; float maxDepth;
    float4* records;      // [N*3] by rank
    unsigned* new_flags;  // [N] by rank
    unsigned* owner;      // [max_surfels], 0xFFFFFFFF = none
};

// one channel of tex4_linear (same expression), fed from staged texels
__device__ __forceinline__ float bilerp(float a, float b, float c, float d, float wx, float wy)
{
    const float top = a * (1.0f - wx) + b * wx, bot = c * (1.0f - wx) + d * wx;
    return top * (1.0f - wy) + bot * wy;
}

// Four lanes per pixel.  A lane alone walks 16 window samples, each a chain of IEEE sqrt / divisions / acos (about
// 4400 VALU instructions per wave with only ~1 wave per SIMD in flight: 30 us of single-wave latency, measured).
// Here the quad (a) stages the 4x4 texel neighbourhood of the window once -- index, vertConf.xyz, normRad.xyz, one
// patch row per lane, all loads independent -- in LDS, and (b) splits the OUTER window loop: lane s evaluates the
// outer iterations s, s+4, ... with the shader's own f32 loop counters, then the quad reduces to the candidate the
// sequential loop would have kept (smallest distance, earliest on ties).  Lane 0 writes the record.
static constexpr int kAssocItems = kB / 4;   // pixels per workgroup
static constexpr int kPatchStride = 7 * 16 + 1;  // words per staged neighbourhood (+1: odd stride, no bank conflicts)

__global__ void __launch_bounds__(kB) associate_kernel(const Batch<FuseArgs> B, int xcd)
{
    __shared__ float s_patch[kAssocItems * kPatchStride];
    const VBlock vb = batch_decode(B.h);
    const FuseArgs& a = B.m[vb.model];
    const int cols = a.cols, rows = a.rows;
    // Only pixels whose integer coordinates share the parity of `time` survive the shader's first test
    // ((int)x == i, (int)y == j: the texcoords are pixel centres), so the launch enumerates just that quarter of
    // the image; new_flags is cleared by a memset beforehand.
    const int par = a.time % 2;
    const int n_i = (cols - par + 1) / 2, n_j = (rows - par + 1) / 2;
    const int gt = xcd_block(xcd, vb.bid, vb.nblk) * kB + threadIdx.x;
    const int t = gt >> 2, sub = gt & 3;
    float* const P = s_patch + (threadIdx.x >> 2) * kPatchStride;  // word c of texel k: P[c * 16 + k]
    bool alive = t < n_i * n_j;
    int i = 0, j = 0, q = 0, rank = 0;
    float tcx = 0, tcy = 0, x = 0, y = 0;
    f3 vPosLocal{0, 0, 0};
    if (alive) {
        j = 2 * (t / n_i) + par; i = 2 * (t % n_i) + par;
        q = j * cols + i; rank = i * rows + j;
        tcx = a.tcx[i]; tcy = a.tcy[j];
        x = tcx * (float)cols; y = tcy * (float)rows;
        alive = (((int)x % 2 == a.time % 2) && ((int)y % 2 == a.time % 2)) && ((int)a.mask[q] == a.maskID);
    }
    const float* dr = a.depth_raw;
    if (alive)
        alive = !(dr[j * cols + iclamp(i - 1, 0, cols - 1)] == 0 || dr[iclamp(j - 1, 0, rows - 1) * cols + i] == 0 ||
                  dr[j * cols + iclamp(i + 1, 0, cols - 1)] == 0 || dr[iclamp(j + 1, 0, rows - 1) * cols + i] == 0);
    const float cx = a.cam.cx, cy = a.cam.cy;
    if (alive) {
        vPosLocal = get_vertex(dr, cols, rows, i, j, x, y, cx, cy, a.inv_fx, a.inv_fy);
        alive = (vPosLocal.z > 0 && vPosLocal.z <= a.maxDepth);
    }
    const float scale = 1.0f;  // ModelProjection::FACTOR
    const float indexXStep = (1.0f / ((float)cols * scale)) * 0.5f, indexYStep = (1.0f / ((float)rows * scale)) * 0.5f;
    const float windowMultiplier = 2;
    const float iBeg = tcx - (scale * indexXStep * windowMultiplier), jBeg = tcy - (scale * indexYStep * windowMultiplier);
    const int X0 = (int)floorf(iBeg * (float)cols - 0.5f), Y0 = (int)floorf(jBeg * (float)rows - 0.5f);
    if (alive) {  // patch row `sub`
#pragma unroll
        for (int c = 0; c < 4; c++) {
            const int k = sub * 4 + c;
            const int g = iclamp(Y0 + sub, 0, rows - 1) * cols + iclamp(X0 + c, 0, cols - 1);
            const float4 vcf = a.vertConf[g];
            const float4 nrd = a.normRad[g];
            const unsigned idx = a.index[g];
            P[0 * 16 + k] = vcf.x; P[1 * 16 + k] = vcf.y; P[2 * 16 + k] = vcf.z;
            P[3 * 16 + k] = nrd.x; P[4 * 16 + k] = nrd.y; P[5 * 16 + k] = nrd.z;
            P[6 * 16 + k] = __uint_as_float(idx);
        }
    }
    __syncthreads();
    if (!alive) return;  // the four lanes of a pixel take the same decision

    const f3 vPos = xform_point(a.pose, vPosLocal);
    const f3 vPos_f = get_vertex(a.depth_filt, cols, rows, i, j, x, y, cx, cy, a.inv_fx, a.inv_fy);
    const f3 vNormLocal = get_normal_central(vPos_f, a.depth_filt, cols, rows, i, j, x, y, cx, cy, a.inv_fx, a.inv_fy);
    const float xl = (x - cx) * a.inv_fx, yl = (y - cy) * a.inv_fy;
    const float lambda = sqrtf(xl * xl + yl * yl + 1);
    const f3 ray = {xl, yl, 1};
    float bestDist = 1000;
    unsigned best = 0; int operation = 0, bestSeq = 0;
    // lane `sub` owns the outer iterations sub, sub+4, ...: it reaches its counter value by the same sequence of
    // f32 additions as the shader's loop and tests the same bound (the counter is monotonic, so the test of its
    // own value decides whether the iteration exists); the four lanes then run their inner loops simultaneously
    float ii = iBeg;
    for (int s4 = 0; s4 < sub; s4++) ii += indexXStep;
    for (int outer = sub; ii < tcx + (scale * indexXStep * windowMultiplier);
         outer += 4, ii += indexXStep, ii += indexXStep, ii += indexXStep, ii += indexXStep) {
        int inner = 0;
        for (float jj = jBeg; jj < tcy + (scale * indexYStep * windowMultiplier); jj += indexYStep, inner++) {
            const int nlx = (int)floorf(ii * (float)cols) - X0, nly = (int)floorf(jj * (float)rows) - Y0;
            const unsigned current = ((unsigned)nlx < 4u && (unsigned)nly < 4u)
                                         ? __float_as_uint(P[6 * 16 + nly * 4 + nlx])
                                         : a.index[nearest_texel(jj, rows) * cols + nearest_texel(ii, cols)];
            if (current > 0U) {
                const float fu = ii * (float)cols - 0.5f, fv = jj * (float)rows - 0.5f;
                const float x0f = floorf(fu), y0f = floorf(fv);
                const int lx = (int)x0f - X0, ly = (int)y0f - Y0;
                const bool staged = (unsigned)lx < 3u && (unsigned)ly < 3u;
                const float wx = fu - x0f, wy = fv - y0f;
                const float* qd = P + (ly * 4 + lx);
                float4 vertConf;
                if (staged) {
                    vertConf.x = bilerp(qd[0 * 16], qd[0 * 16 + 1], qd[0 * 16 + 4], qd[0 * 16 + 5], wx, wy);
                    vertConf.y = bilerp(qd[1 * 16], qd[1 * 16 + 1], qd[1 * 16 + 4], qd[1 * 16 + 5], wx, wy);
                    vertConf.z = bilerp(qd[2 * 16], qd[2 * 16 + 1], qd[2 * 16 + 4], qd[2 * 16 + 5], wx, wy);
                } else
                    vertConf = tex4_linear(a.vertConf, cols, rows, ii, jj);
                const float zdiff = (vertConf.z - vPosLocal.z);
                if (fabsf(zdiff * lambda) < 0.05f) {
                    const float dist = norm(cross(ray, f3{vertConf.x, vertConf.y, vertConf.z}));
                    float4 normRad;
                    if (staged) {
                        normRad.x = bilerp(qd[3 * 16], qd[3 * 16 + 1], qd[3 * 16 + 4], qd[3 * 16 + 5], wx, wy);
                        normRad.y = bilerp(qd[4 * 16], qd[4 * 16 + 1], qd[4 * 16 + 4], qd[4 * 16 + 5], wx, wy);
                        normRad.z = bilerp(qd[5 * 16], qd[5 * 16 + 1], qd[5 * 16 + 4], qd[5 * 16 + 5], wx, wy);
                    } else
                        normRad = tex4_linear(a.normRad, cols, rows, ii, jj);
                    if (dist < bestDist && (fabsf(normRad.z) < 0.75f || fabsf(angle_between(f3{normRad.x, normRad.y, normRad.z}, vNormLocal)) < 0.5f)) {
                        operation = 1; bestDist = dist; best = current; bestSeq = outer * 64 + inner;
                    }
                }
            }
        }
    }
    // quad reduction to the sequential loop's survivor: smallest distance, earliest (outer, inner) on ties
#pragma unroll
    for (int o = 1; o <= 2; o <<= 1) {
        const int oOp = __shfl_xor(operation, o, 4), oSeq = __shfl_xor(bestSeq, o, 4);
        const float oDist = __shfl_xor(bestDist, o, 4);
        const unsigned oBest = (unsigned)__shfl_xor((int)best, o, 4);
        const bool take = oOp && (!operation || oDist < bestDist || (oDist == bestDist && oSeq < bestSeq));
        if (take) { operation = 1; bestDist = oDist; best = oBest; bestSeq = oSeq; }
    }
    if (sub != 0) return;
    const uchar4 c = a.rgba[q];
    const f3 nG = xform_dir(a.pose, vNormLocal);
    const float radius = get_radius(vPos_f.z, vNormLocal.z, a.inv_fx, a.inv_fy);
    const float conf = confidence(x, y, cx, cy, a.weighting);
    const float col = (float)(((int)c.x << 16) + ((int)c.y << 8) + (int)c.z);
    a.records[rank * 3 + 0] = make_float4(vPos.x, vPos.y, vPos.z, conf);
    a.records[rank * 3 + 1] = make_float4(col, 0.f, (float)a.time, operation == 1 ? -1.f : -2.f);
    a.records[rank * 3 + 2] = make_float4(nG.x, nG.y, nG.z, radius);
    if (operation == 1) atomicMin(&a.owner[best], (unsigned)rank);
    else a.new_flags[rank] = 1;
}

// update.vert:38-111 (reads the winning record through the owner index; resets the owner slot)
struct UpdateArgs { const float4* in; const unsigned* count; unsigned* owner; const float4* records; int time; float4* out; };
__device__ __forceinline__ void update_body(const Batch<UpdateArgs>& B, int blk)
{
    const VBlock vb = batch_decode_at(B.h, blk);
    const UpdateArgs& ua = B.m[vb.model];
    const float4* __restrict__ in = ua.in; unsigned* __restrict__ owner = ua.owner; const float4* __restrict__ records = ua.records;
    float4* __restrict__ out = ua.out; const int time = ua.time;
    const unsigned id = vb.bid * kB + threadIdx.x;
    if (id >= *ua.count) return;
    const float4 pc = in[id * 3], ct = in[id * 3 + 1], nr = in[id * 3 + 2];
    const unsigned ow = owner[id];
    if (ow == 0xFFFFFFFFu) { out[id * 3] = pc; out[id * 3 + 1] = ct; out[id * 3 + 2] = nr; return; }
    owner[id] = 0xFFFFFFFFu;
    const float4 rp = records[ow * 3], rc = records[ow * 3 + 1], rn = records[ow * 3 + 2];
    const float c_k = pc.w, a = rp.w;
    if (rn.w < (1.0f + 0.5f) * nr.w) {
        float4 op, oc, on;
        op.x = ((c_k * pc.x) + (a * rp.x)) / (c_k + a);
        op.y = ((c_k * pc.y) + (a * rp.y)) / (c_k + a);
        op.z = ((c_k * pc.z) + (a * rp.z)) / (c_k + a);
        op.w = c_k + a;
        const f3 oldc = decode_color(ct.x), newc = decode_color(rc.x);
        oc.x = encode_color(((c_k * oldc.x) + (a * newc.x)) / (c_k + a), ((c_k * oldc.y) + (a * newc.y)) / (c_k + a),
                            ((c_k * oldc.z) + (a * newc.z)) / (c_k + a));
        oc.y = ct.y; oc.z = ct.z; oc.w = (float)time;
        const float n0 = ((c_k * nr.x) + (a * rn.x)) / (c_k + a), n1 = ((c_k * nr.y) + (a * rn.y)) / (c_k + a);
        const float n2 = ((c_k * nr.z) + (a * rn.z)) / (c_k + a), n3 = ((c_k * nr.w) + (a * rn.w)) / (c_k + a);
        const f3 nn = normalized(f3{n0, n1, n2});
        on = make_float4(nn.x, nn.y, nn.z, n3);
        out[id * 3] = op; out[id * 3 + 1] = oc; out[id * 3 + 2] = on;
    } else {
        out[id * 3] = make_float4(pc.x, pc.y, pc.z, c_k + a);
        out[id * 3 + 1] = make_float4(ct.x, ct.y, ct.z, (float)time);
        out[id * 3 + 2] = nr;
    }
}
__global__ void __launch_bounds__(kB) update_kernel(const Batch<UpdateArgs> B) { update_body(B, (int)blockIdx.x); }

// Launches that do not depend on each other, side by side in ONE grid (late in round 6).  Behind the association the frame's chain was
// block sums -> compaction -> update -> index keys -> ...: the compaction of the new surfels (flags / records of the association -> `fresh`)
// and the update of the old ones (-> the other surfel buffer) never read what the other writes, and neither does the second index pass's
// rasterisation, which only needs the update.  So: {update || block sums}, then {index keys || compaction} -- two launches of 5-6 us and their
// boundaries off the chain.  The longer part comes first in the grid.
static_assert(sizeof(Batch<IndexArgs>) + sizeof(FrameGeom) + sizeof(Batch<ScanArgs>) + 16 <= 4096 && sizeof(Batch<UpdateArgs>) + sizeof(Batch<ScanArgs>) + 16 <= 4096,
              "the side-by-side launches carry two batches' arguments in one 4 KB kernel-argument segment");
__global__ void __launch_bounds__(kB) update_blocksums_kernel(const Batch<UpdateArgs> U, const Batch<ScanArgs> S, int u_blocks)
{
    const int b = (int)blockIdx.x;
    if (b < u_blocks) update_body(U, b);
    else scan_block_sums_body(S, b - u_blocks);
}
__global__ void __launch_bounds__(kB) indexsplat_scatter_kernel(const Batch<IndexArgs> I, const FrameGeom g, const Batch<ScanArgs> S, int i_blocks)
{
    const int b = (int)blockIdx.x;
    if (b < i_blocks) index_splat_body(I, g, b);
    else scan_scatter_body(S, b - i_blocks);
}

// ====================================================================================== clean ====
// copy_unstable.vert:53-149 (deformation-graph branch dead: nodes == 0).  The tested/modified surfel
// is written to `staged` at its input position together with its keep flag; an ordered scan + scatter
// then reproduces the transform-feedback output order (old surfels first, appended ones after).
struct CleanArgs {
    const unsigned* index; const float4* vertConf; const float4* colorTime;
    const float* depth_filt; const unsigned char* mask;
    Mat4 t_inv; cf_cam cam; int cols, rows, time; float confThreshold, outlierCoeff; int timeDelta, maskID;
    const float4* rec;   // nullable: [rows * cols][2] vertConf | colorTime.zw, index, filtered depth per texel, packed by the index pass in front of this one (IndexArgs::clean_rec)
    // (the launch's per-model buffers, kernel parameters until round 4)
    const float4* surfels; const unsigned* count; const float4* fresh; const unsigned* n_fresh; unsigned total_bound; float4* staged; unsigned* flags;
    int xcd;   // workgroup order of THIS model's share (xcd_block): runs of 64 per XCD for a large map, the hardware's order for a small one
#ifdef CF_ABLATE
    int abl;   // diagnostics build (CF_CLEAN_ABLATE): 1 no staging of the texel patch, 2 no 4x4 window, 4 no 3x3 depth window -- timing only, results are wrong
#endif
};

// The 4x4 half-pixel window touches at most a 4x4 texel neighbourhood of the index map textures.  Reading it
// sample by sample (16 x (1 + 2x4) gathers per surfel) thrashes L1 and the XCD's L2 (measured: 560 MB of fabric
// reads per launch for 275 k surfels).  As in associate_kernel, four lanes share a surfel: the quad stages the
// neighbourhood -- vertConf.xyzw, colorTime.zw and the index, one patch row per lane -- in LDS with independent
// loads, splits the outer window loop (lane s takes iterations s, s+4, ...; the shader's f32 loop counters are
// kept), and adds up the two vote counts, which do not depend on the order.  Texels outside the staged patch
// (only reachable through the f32 loop-counter corner cases) fall back to the global fetch.
// Late in round 6 ("measure the rewrite"): timing ablations of this kernel (diagnostics build, CF_CLEAN_ABLATE;
// profiles/r6zd_*) put its 70 us at 17 us for loading / projecting / storing the surfels, 24 us for STAGING the neighbourhood -- twelve loads
// per lane from three arrays, every lane of a wave on a row of its own: 64 segments per instruction --, 14 us for the 4x4 window's
// arithmetic and 12 us for the 3x3 depth window's nine gathers per lane.  So the staging reads ONE array now, packed by the index pass in
// front of this one (index_resolve_kernel: vertConf | colorTime.zw, index, filtered depth = 32 bytes per texel), in 16-byte pieces dealt to the
// quad's lanes so that a quad reads 64 consecutive bytes per instruction (eight loads per lane, 16 segments per instruction), and the depth
// window is served from the staged patch (an eighth word per texel).  Without the records (cf_model_clean called alone) the old staging runs.
// (Measured as well: only colorTime.zw | index | depth in the record, 16 bytes, vertConf staged from its own array -- the index pass cheaper
// by 2.4 us, this kernel 64.9 against 52.2 us.)
// The eighth word makes a 256-thread workgroup's patches 33 KB: four workgroups per CU where 29 KB allowed five, and the kernel lives on its
// occupancy (81 against 70 us for the old staging with the larger stride).  Workgroups of kCleanB threads (8.3 KB each at 64) fill the
// CU's 160 KB to within a wave of the old occupancy.
#ifndef CF_CLEAN_B
#define CF_CLEAN_B 64
#endif
static constexpr int kCleanB = CF_CLEAN_B;
static constexpr int kCleanItems = kCleanB / 4;
static constexpr int kCleanStride = 8 * 16 + 1;  // words per staged neighbourhood: 8 words x 16 texels (+1: odd stride)

__global__ void __launch_bounds__(kCleanB) clean_kernel(const Batch<CleanArgs> B)
{
    __shared__ float s_patch[kCleanItems * kCleanStride];
    const VBlock vb = batch_decode(B.h);
    const CleanArgs& a = B.m[vb.model];
    const float4* __restrict__ surfels = a.surfels; const unsigned* __restrict__ count = a.count; const float4* __restrict__ fresh = a.fresh;
    const unsigned* __restrict__ n_fresh = a.n_fresh; const unsigned total_bound = a.total_bound; float4* __restrict__ staged = a.staged;
    unsigned* __restrict__ flags = a.flags;
    float* const P = s_patch + (threadIdx.x >> 2) * kCleanStride;  // word c of texel t: P[c * 16 + t]
    const float4* __restrict__ const rec = a.rec;
    const unsigned gt = (unsigned)xcd_block(a.xcd, vb.bid, vb.nblk) * kCleanB + threadIdx.x;
    const unsigned k = gt >> 2;
    const int sub = (int)(gt & 3u);
    const unsigned n_old = *count, n_all = n_old + *n_fresh;
    const bool exists = k < n_all;
    if (!exists && sub == 0 && k < total_bound) flags[k] = 0;
    const int cols = a.cols, rows = a.rows;
    const float scale = 1.0f;
    float4 pc = make_float4(0, 0, 0, 0), ct = pc, nr = pc;
    f3 localPos{0, 0, 0}, localNorm{0, 0, 0};
    float x = 0, y = 0, x_n = 0, y_n = 0;
    const float stepX = 1.0f / (float)cols, stepY = 1.0f / (float)rows;
    const float indexXStep = stepX * 0.5f / scale, indexYStep = stepY * 0.5f / scale;
    const float windowMultiplier = 2;
    bool window = false;
    if (exists) {
        const float4* src = (k < n_old) ? surfels + (size_t)k * 3 : fresh + (size_t)(k - n_old) * 3;
        pc = src[0]; ct = src[1]; nr = src[2];
        localPos = xform_point(a.t_inv, f3{pc.x, pc.y, pc.z});
        x = ((a.cam.fx * localPos.x) / localPos.z) + a.cam.cx; y = ((a.cam.fy * localPos.y) / localPos.z) + a.cam.cy;
        localNorm = normalized(xform_dir(a.t_inv, f3{nr.x, nr.y, nr.z}));
        x_n = x / (float)cols; y_n = y / (float)rows;
        window = (float)a.time - ct.w < (float)a.timeDelta && localPos.z > 0 && x > 0 && y > 0 && x < (float)cols && y < (float)rows;
    }
    const float iBeg = x_n - (scale * indexXStep * windowMultiplier), jBeg = y_n - (scale * indexYStep * windowMultiplier);
    const int X0 = (int)floorf(iBeg * (float)cols - 0.5f), Y0 = (int)floorf(jBeg * (float)rows - 0.5f);
#ifdef CF_ABLATE
    const int abl = a.abl;
#else
    constexpr int abl = 0;
#endif
    if (window && rec && !(abl & 1)) {  // the quad's 32 pieces of 16 bytes (4 rows x 4 texels x 2 halves): lane `sub` takes pieces sub, sub + 4, ...
        const int half = sub & 1;
#pragma unroll
        for (int q = 0; q < 8; q++) {
            const int row = q >> 1, tex = (sub >> 1) + 2 * (q & 1), t = row * 4 + tex;
            const int g = iclamp(Y0 + row, 0, rows - 1) * cols + iclamp(X0 + tex, 0, cols - 1);
            const float4 v = rec[2 * g + half];
            float* const o = P + half * 64 + t;   // words 0..3 (vertConf) or 4..7 (colorTime.z, .w, index, depth)
            o[0] = v.x; o[16] = v.y; o[32] = v.z; o[48] = v.w;
        }
    } else
    if (window && !(abl & 1)) {  // patch row `sub`
#pragma unroll
        for (int c = 0; c < 4; c++) {
            const int t = sub * 4 + c;
            const int g = iclamp(Y0 + sub, 0, rows - 1) * cols + iclamp(X0 + c, 0, cols - 1);
            const float4 vcf = a.vertConf[g];
            const float4 ctm = a.colorTime[g];
            const unsigned idx = a.index[g];
            P[0 * 16 + t] = vcf.x; P[1 * 16 + t] = vcf.y; P[2 * 16 + t] = vcf.z; P[3 * 16 + t] = vcf.w;
            P[4 * 16 + t] = ctm.z; P[5 * 16 + t] = ctm.w;
            P[6 * 16 + t] = __uint_as_float(idx);
        }
    }
    __syncthreads();
    if (!exists) return;
    int test = 1;
    int cnt = 0, zCount = 0, violationCount = 0;
    float avgViolation = 0;
    if (window && !(abl & 2)) {
        // lane `sub` owns the outer iterations sub, sub+4, ... (see associate_kernel)
        float i = iBeg;
        for (int s4 = 0; s4 < sub; s4++) i += indexXStep;
        for (; i < x_n + (scale * indexXStep * windowMultiplier); i += indexXStep, i += indexXStep, i += indexXStep, i += indexXStep) {
            for (float j = jBeg; j < y_n + (scale * indexYStep * windowMultiplier); j += indexYStep) {
                const int nlx = (int)floorf(i * (float)cols) - X0, nly = (int)floorf(j * (float)rows) - Y0;
                const unsigned current = ((unsigned)nlx < 4u && (unsigned)nly < 4u)
                                             ? __float_as_uint(P[6 * 16 + nly * 4 + nlx])
                                             : a.index[nearest_texel(j, rows) * cols + nearest_texel(i, cols)];
                if (current > 0U) {
                    float4 vertConf, colorTime;
                    const float fu = i * (float)cols - 0.5f, fv = j * (float)rows - 0.5f;
                    const float x0f = floorf(fu), y0f = floorf(fv);
                    const int lx = (int)x0f - X0, ly = (int)y0f - Y0;
                    if ((unsigned)lx < 3u && (unsigned)ly < 3u) {
                        const float wx = fu - x0f, wy = fv - y0f;
                        const float* q = P + (ly * 4 + lx);
                        vertConf.x = bilerp(q[0 * 16], q[0 * 16 + 1], q[0 * 16 + 4], q[0 * 16 + 5], wx, wy);
                        vertConf.y = bilerp(q[1 * 16], q[1 * 16 + 1], q[1 * 16 + 4], q[1 * 16 + 5], wx, wy);
                        vertConf.z = bilerp(q[2 * 16], q[2 * 16 + 1], q[2 * 16 + 4], q[2 * 16 + 5], wx, wy);
                        vertConf.w = bilerp(q[3 * 16], q[3 * 16 + 1], q[3 * 16 + 4], q[3 * 16 + 5], wx, wy);
                        colorTime.z = bilerp(q[4 * 16], q[4 * 16 + 1], q[4 * 16 + 4], q[4 * 16 + 5], wx, wy);
                        colorTime.w = bilerp(q[5 * 16], q[5 * 16 + 1], q[5 * 16 + 4], q[5 * 16 + 5], wx, wy);
                    } else {
                        vertConf = tex4_linear(a.vertConf, cols, rows, i, j);
                        colorTime = tex4_linear(a.colorTime, cols, rows, i, j);
                    }
                    const float dx = vertConf.x - localPos.x, dy = vertConf.y - localPos.y;
                    if (colorTime.z < ct.z && vertConf.w > a.confThreshold && vertConf.z > localPos.z && vertConf.z - localPos.z < 0.01f &&
                        sqrtf(dx * dx + dy * dy) < nr.w * 1.4f)
                        cnt++;
                    if (colorTime.w == (float)a.time && vertConf.w > a.confThreshold && vertConf.z > localPos.z &&
                        vertConf.z - localPos.z > 0.01f && fabsf(localNorm.z) > 0.85f)
                        zCount++;
                }
            }
        }
    }
    if (window && !(abl & 4)) {
        // every lane of the quad walks the 3x3 depth window (sequential f32 sum) so that all four hold the result
        for (float i = x_n - stepX; i <= x_n + stepX; i += stepX)
            for (float j = y_n - stepY; j <= y_n + stepY; j += stepY) {
                const int tx = nearest_texel(i, cols), ty = nearest_texel(j, rows);
                const int lx = tx - X0, ly = ty - Y0;   // (inside the staged patch unless the f32 loop counters stray)
                const float dv = (rec && (unsigned)lx < 4u && (unsigned)ly < 4u) ? P[7 * 16 + ly * 4 + lx] : a.depth_filt[ty * cols + tx];
                const float d = dv - localPos.z;
                if (d > 0.03f) { violationCount++; avgViolation += d; }
            }
    }
    cnt += __shfl_xor(cnt, 1, 4); cnt += __shfl_xor(cnt, 2, 4);
    zCount += __shfl_xor(zCount, 1, 4); zCount += __shfl_xor(zCount, 2, 4);
    if (sub != 0) return;
    if (cnt > 8 || zCount > 4) test = 0;
    if (ct.w == -2) ct.w = (float)a.time;
    if ((ct.w == -1 || (((float)a.time - ct.w) > 20 && pc.w < a.confThreshold))) test = 0;
    if (ct.w > 0 && (float)a.time - ct.w > (float)a.timeDelta) test = 1;
    if (violationCount > 0) {
        avgViolation /= (float)violationCount;
        pc.w *= 1.0f / (1 + a.outlierCoeff * avgViolation);
        const int mx = nearest_texel(x_n, cols), my = nearest_texel(y_n, rows);
        const int maskValue = a.mask[my * cols + mx];
        const float wDepth = a.depth_filt[my * cols + mx];
        if (maskValue != a.maskID && (wDepth > localPos.z - 0.05f && wDepth < localPos.z + 0.05f))
            pc.w *= (0.5f + 0.5f * (1 - a.outlierCoeff / 10.0f));
    }
    flags[k] = (unsigned)test;
    staged[(size_t)k * 3] = pc; staged[(size_t)k * 3 + 1] = ct; staged[(size_t)k * 3 + 2] = nr;
}

// zero-fill of one u32 buffer per model (the new-surfel flags before association: one launch instead of a memset per model)
struct FillArgs { uint4* p; long long n16; };   // n16: 16-byte words
__global__ void __launch_bounds__(kB) fill_zero_kernel(const Batch<FillArgs> B)
{
    const VBlock vb = batch_decode(B.h);
    const FillArgs& a = B.m[vb.model];
    const long long i = (long long)vb.bid * kB + threadIdx.x;
    if (i < a.n16) a.p[i] = make_uint4(0, 0, 0, 0);
}
__global__ void set_count_kernel(unsigned* out, unsigned v) { *out = v; }
__global__ void set_count2_kernel(unsigned* out, unsigned* out_host, unsigned v) { *out = v; if (out_host) *out_host = v; }

// ------------------------------------------------------------------------------- launchers ----
void launch_bilateral(hipStream_t s, const float* depth, int cols, int rows, float maxD, float* out)
{
    bilateral_kernel<<<gridFor((long long)cols * rows), kB, 0, s>>>(depth, cols, rows, maxD, out);
}

static Mat4 mat4_from(const float m[16]) { Mat4 r; for (int i = 0; i < 16; i++) r.m[i] = m[i]; return r; }

void launch_feedback(hipStream_t s, const uint8_t* rgba, const float* depth, int cols, int rows, cf_cam cam, float inv_fx, float inv_fy,
                     const float* tcx, const float* tcy, int time, float maxDepth, float* rec, unsigned* flags)
{
    feedback_kernel<<<gridFor((long long)cols * rows), kB, 0, s>>>(reinterpret_cast<const uchar4*>(rgba), depth, cols, rows, cam.cx, cam.cy, inv_fx,
                                                                   inv_fy, tcx, tcy, time, maxDepth, reinterpret_cast<float4*>(rec), flags);
}
void launch_init(hipStream_t s, const float* raw, const float* filt, const unsigned* raw_count, long long max_n, float* out)
{
    init_kernel<<<gridFor(max_n), kB, 0, s>>>(reinterpret_cast<const float4*>(raw), reinterpret_cast<const float4*>(filt), raw_count,
                                              reinterpret_cast<float4*>(out));
}
// the arguments of a batched launch travel in the 4 KB kernel-argument segment
static_assert(sizeof(Batch<FuseArgs>) + 16 <= 4096 && sizeof(Batch<CleanArgs>) + 16 <= 4096 && sizeof(Batch<IndexArgs>) + sizeof(FrameGeom) <= 4096 &&
              sizeof(Batch<SplatArgs>) + sizeof(FrameGeom) <= 4096 && sizeof(Batch<ScanArgs>) <= 4096, "lower kSurfBatch");
// ---- predictIndices -------------------------------------------------------------------------------------------------------------
static IndexArgs index_args(const IndexPassArgs& h)
{
    return IndexArgs{reinterpret_cast<const float4*>(h.surfels), h.count, mat4_from(h.t_inv), h.maxDepth, h.time, h.timeDelta, h.id_begin, h.id_end,
                     h.keys, h.index, reinterpret_cast<float4*>(h.vertConf), reinterpret_cast<float4*>(h.colorTime), reinterpret_cast<float4*>(h.normRad),
                     h.t_inv_dev, reinterpret_cast<float4*>(h.clean_rec), h.clean_depth};
}
void launch_index_keys_batch(hipStream_t s, const IndexPassArgs* items, int n_items, cf_cam cam, int cols, int rows)
{
    const FrameGeom g{cam, cols, rows};
    for (int base = 0; base < n_items; base += kSurfBatch) {
        const int nb = n_items - base < kSurfBatch ? n_items - base : kSurfBatch;
        Batch<IndexArgs> B;
        int blocks[kSurfBatch], any = 0;
        for (int k = 0; k < nb; k++) {
            B.m[k] = index_args(items[base + k]);
            blocks[k] = B.m[k].id_end > B.m[k].id_begin ? gridFor(B.m[k].id_end - B.m[k].id_begin) : 0;
            any += blocks[k];
        }
        if (any) index_splat_kernel<<<batch_layout(B, blocks, nb), kB, 0, s>>>(B, g);
    }
}
void launch_index_resolve_batch(hipStream_t s, const IndexPassArgs* items, int n_items, cf_cam cam, int cols, int rows)
{
    const FrameGeom g{cam, cols, rows};
    for (int base = 0; base < n_items; base += kSurfBatch) {
        const int nb = n_items - base < kSurfBatch ? n_items - base : kSurfBatch;
        Batch<IndexArgs> B;
        int blocks[kSurfBatch];
        for (int k = 0; k < nb; k++) { B.m[k] = index_args(items[base + k]); blocks[k] = gridFor((long long)cols * rows); }
        index_resolve_kernel<<<batch_layout(B, blocks, nb), kB, 0, s>>>(B, g);
    }
}
void launch_index_keys(hipStream_t s, const float* surfels, const unsigned* count, unsigned id_begin, unsigned id_end, const float t_inv[16],
                       cf_cam cam, int cols, int rows, float maxDepth, int time, int timeDelta, unsigned long long* keys)
{
    IndexPassArgs a{};
    a.surfels = surfels; a.count = count; a.id_begin = id_begin; a.id_end = id_end; for (int q = 0; q < 16; q++) a.t_inv[q] = t_inv[q];
    a.maxDepth = maxDepth; a.time = time; a.timeDelta = timeDelta; a.keys = keys;
    launch_index_keys_batch(s, &a, 1, cam, cols, rows);
}
void launch_index_resolve(hipStream_t s, const float* surfels, const float t_inv[16], int cols, int rows, unsigned long long* keys,
                          unsigned* index, float* vertConf, float* colorTime, float* normRad)
{
    IndexPassArgs a{};
    a.surfels = surfels; for (int q = 0; q < 16; q++) a.t_inv[q] = t_inv[q]; a.keys = keys; a.index = index; a.vertConf = vertConf; a.colorTime = colorTime;
    a.normRad = normRad;
    launch_index_resolve_batch(s, &a, 1, cf_cam{}, cols, rows);
}
void launch_predict_indices(hipStream_t s, const float* surfels, const unsigned* count, unsigned count_bound, const float t_inv[16], cf_cam cam,
                            int cols, int rows, float maxDepth, int time, int timeDelta, unsigned long long* keys, unsigned* index,
                            float* vertConf, float* colorTime, float* normRad)
{
    launch_index_keys(s, surfels, count, 0, count_bound, t_inv, cam, cols, rows, maxDepth, time, timeDelta, keys);
    launch_index_resolve(s, surfels, t_inv, cols, rows, keys, index, vertConf, colorTime, normRad);
}
// ---- combinedPredict ------------------------------------------------------------------------------------------------------------
void launch_combined_predict_batch(hipStream_t s, const SplatPassArgs* items, int n_items, cf_cam cam, int cols, int rows)
{
    const FrameGeom g{cam, cols, rows};
    for (int base = 0; base < n_items; base += kSurfBatch) {
        const int nb = n_items - base < kSurfBatch ? n_items - base : kSurfBatch;
        Batch<SplatArgs> B;
        int blocks[kSurfBatch], any = 0;
        for (int k = 0; k < nb; k++) {
            const SplatPassArgs& h = items[base + k];
            B.m[k] = SplatArgs{reinterpret_cast<const float4*>(h.surfels), h.count, mat4_from(h.t_inv), h.maxDepth, h.confThreshold, h.time, h.maxTime,
                               h.timeDelta, reinterpret_cast<const float4*>(h.rays), h.keys, reinterpret_cast<uchar4*>(h.image),
                               reinterpret_cast<float4*>(h.vertexConf), reinterpret_cast<float4*>(h.normalRad), h.time16};
            blocks[k] = h.count_bound > 0 ? gridFor(4ll * h.count_bound) : 0;
            any += blocks[k];
        }
        if (any) splat_raster_kernel<<<batch_layout(B, blocks, nb), kB, 0, s>>>(B, g);
        for (int k = 0; k < nb; k++) blocks[k] = gridFor((long long)cols * rows);
        splat_resolve_kernel<<<batch_layout(B, blocks, nb), kB, 0, s>>>(B, g);
    }
}
void launch_combined_predict(hipStream_t s, const float* surfels, const unsigned* count, unsigned count_bound, const float t_inv[16], cf_cam cam,
                             int cols, int rows, float maxDepth, float confThreshold, int time, int maxTime, int timeDelta,
                             const float* rays, unsigned long long* keys, uint8_t* image, float* vertexConf, float* normalRad,
                             uint16_t* time16)
{
    SplatPassArgs a{};
    a.surfels = surfels; a.count = count; a.count_bound = count_bound; for (int q = 0; q < 16; q++) a.t_inv[q] = t_inv[q]; a.maxDepth = maxDepth;
    a.confThreshold = confThreshold; a.time = time; a.maxTime = maxTime; a.timeDelta = timeDelta; a.rays = rays; a.keys = keys; a.image = image;
    a.vertexConf = vertexConf; a.normalRad = normalRad; a.time16 = time16;
    launch_combined_predict_batch(s, &a, 1, cam, cols, rows);
}
void launch_splat_rays(hipStream_t s, cf_cam cam, int cols, int rows, float* rays)
{
    splat_rays_kernel<<<gridFor((long long)cols * rows), kB, 0, s>>>(cam, cols, rows, reinterpret_cast<float4*>(rays));
}
void launch_fill_in(hipStream_t s, const float* pv, const float* pn, const uint8_t* pimg, const float* depth, const uint8_t* rgba, int cols,
                    int rows, cf_cam cam, float inv_fx, float inv_fy, int pass_geom, int pass_rgb, float* ov, float* on, uint8_t* oi)
{
    fill_in_kernel<<<gridFor((long long)cols * rows), kB, 0, s>>>(
        reinterpret_cast<const float4*>(pv), reinterpret_cast<const float4*>(pn), reinterpret_cast<const uchar4*>(pimg), depth,
        reinterpret_cast<const uchar4*>(rgba), cols, rows, cam.cx, cam.cy, inv_fx, inv_fy, pass_geom, pass_rgb, reinterpret_cast<float4*>(ov),
        reinterpret_cast<float4*>(on), reinterpret_cast<uchar4*>(oi));
}
void launch_fill_ratio(hipStream_t s, const uint8_t* pimg, int cols, int rows, unsigned* out2, unsigned* out2_host)
{
    fill_ratio_kernel<<<1, 1024, 0, s>>>(reinterpret_cast<const uchar4*>(pimg), cols, rows, out2, out2_host);
}
// ---- fuse: association (+ the flag buffers zeroed in one launch), update ---------------------------------------------------------------
void launch_associate_batch(hipStream_t s, const SurfelFuseArgs* items, int n_items)
{
    const int chunk = xcd_env("CF_XCD_ASSOC", 0);
    for (int base = 0; base < n_items; base += kSurfBatch) {
        const int nb = n_items - base < kSurfBatch ? n_items - base : kSurfBatch;
        Batch<FuseArgs> B;
        Batch<FillArgs> Z;
        int blocks[kSurfBatch], zblocks[kSurfBatch], zany = 0;
        for (int k = 0; k < nb; k++) {
            const SurfelFuseArgs& h = items[base + k];
            FuseArgs& a = B.m[k];
            a.index = h.index; a.vertConf = reinterpret_cast<const float4*>(h.vertConf); a.normRad = reinterpret_cast<const float4*>(h.normRad);
            a.rgba = reinterpret_cast<const uchar4*>(h.rgba); a.depth_raw = h.depth_raw; a.depth_filt = h.depth_filt; a.mask = h.mask;
            a.tcx = h.tcx; a.tcy = h.tcy; a.pose = mat4_from(h.pose); a.cam = h.cam; a.inv_fx = h.inv_fx; a.inv_fy = h.inv_fy;
            a.cols = h.cols; a.rows = h.rows; a.time = h.time; a.weighting = h.weighting; a.maskID = h.maskID; a.maxDepth = h.maxDepth;
            a.records = reinterpret_cast<float4*>(h.records); a.new_flags = h.new_flags; a.owner = h.owner;
            const int par = h.time % 2;
            const long long n = (long long)((h.cols - par + 1) / 2) * ((h.rows - par + 1) / 2);
            blocks[k] = gridXcd(4 * n, chunk);
            const long long n16 = ((long long)h.cols * h.rows + 3) / 4;   // (new_flags holds cols * rows words: cf_model_create rounds it up)
            Z.m[k] = FillArgs{reinterpret_cast<uint4*>(h.new_flags), n16};
            zblocks[k] = h.flags_clean ? 0 : gridFor(n16);   // (left clean by the last compaction: ScanPassArgs::zero_flags)
            zany += zblocks[k];
        }
        if (zany) fill_zero_kernel<<<batch_layout(Z, zblocks, nb), kB, 0, s>>>(Z);
        associate_kernel<<<batch_layout(B, blocks, nb), kB, 0, s>>>(B, chunk);
    }
}
void launch_associate(hipStream_t s, const SurfelFuseArgs& h) { launch_associate_batch(s, &h, 1); }
void launch_update_batch(hipStream_t s, const UpdatePassArgs* items, int n_items)
{
    for (int base = 0; base < n_items; base += kSurfBatch) {
        const int nb = n_items - base < kSurfBatch ? n_items - base : kSurfBatch;
        Batch<UpdateArgs> B;
        int blocks[kSurfBatch], any = 0;
        for (int k = 0; k < nb; k++) {
            const UpdatePassArgs& h = items[base + k];
            B.m[k] = UpdateArgs{reinterpret_cast<const float4*>(h.in), h.count, h.owner, reinterpret_cast<const float4*>(h.records), h.time,
                                reinterpret_cast<float4*>(h.out)};
            blocks[k] = h.count_bound > 0 ? gridFor(h.count_bound) : 0;
            any += blocks[k];
        }
        if (any) update_kernel<<<batch_layout(B, blocks, nb), kB, 0, s>>>(B);
    }
}
void launch_update(hipStream_t s, const float* in, const unsigned* count, unsigned count_bound, unsigned* owner, const float* records, int time,
                   float* out)
{
    const UpdatePassArgs a{in, count, count_bound, owner, records, time, out};
    launch_update_batch(s, &a, 1);
}
// {update || block sums of the new surfels' compaction}, {index keys of the pass behind the update || that compaction} (update_blocksums_kernel):
// what launch_scan_scatter_batch + launch_update_batch + launch_index_keys_batch enqueue as four launches, as two.  false (nothing
// enqueued) when the batch does not fit one launch or a compaction has no elements: the caller then takes the separate launches.
bool launch_update_compaction_index_keys(hipStream_t s, const UpdatePassArgs* up, const ScanPassArgs* sc, const IndexPassArgs* ix, int n, cf_cam cam, int cols,
                                         int rows)
{
    if (n <= 0 || n > kSurfBatch) return false;
    Batch<UpdateArgs> U; Batch<ScanArgs> S; Batch<IndexArgs> I;
    int ub[kSurfBatch], sb[kSurfBatch], ib[kSurfBatch];
    for (int k = 0; k < n; k++) {
        U.m[k] = UpdateArgs{reinterpret_cast<const float4*>(up[k].in), up[k].count, up[k].owner, reinterpret_cast<const float4*>(up[k].records), up[k].time,
                            reinterpret_cast<float4*>(up[k].out)};
        ub[k] = up[k].count_bound > 0 ? gridFor(up[k].count_bound) : 0;
        const ScanPassArgs& h = sc[k];
        S.m[k] = ScanArgs{reinterpret_cast<const float4*>(h.rec), h.flags, h.n, h.block_sums, h.total, h.add_to_total, reinterpret_cast<float4*>(h.out), h.total_host, h.zero_flags};
        sb[k] = (int)((h.n + kScanItems - 1) / kScanItems);
        if (sb[k] == 0) return false;
        I.m[k] = index_args(ix[k]);
        ib[k] = I.m[k].id_end > I.m[k].id_begin ? gridFor(I.m[k].id_end - I.m[k].id_begin) : 0;
    }
    const int ug = batch_layout(U, ub, n), sg = batch_layout(S, sb, n), ig = batch_layout(I, ib, n);
    const FrameGeom g{cam, cols, rows};
    update_blocksums_kernel<<<ug + sg, kB, 0, s>>>(U, S, ug);
    indexsplat_scatter_kernel<<<ig + sg, kB, 0, s>>>(I, g, S, ig);
    return true;
}
// ---- clean ----------------------------------------------------------------------------------------------------------------------
void launch_clean_batch(hipStream_t s, const CleanPassArgs* items, int n_items)
{
    const int chunk = xcd_env("CF_XCD_CLEAN", 64);
    for (int base = 0; base < n_items; base += kSurfBatch) {
        const int nb = n_items - base < kSurfBatch ? n_items - base : kSurfBatch;
        Batch<CleanArgs> B;
        int blocks[kSurfBatch], any = 0;
        for (int k = 0; k < nb; k++) {
            const CleanPassArgs& p = items[base + k];
            const SurfelCleanArgs& h = p.h;
            CleanArgs& a = B.m[k];
            a.index = h.index; a.vertConf = reinterpret_cast<const float4*>(h.vertConf); a.colorTime = reinterpret_cast<const float4*>(h.colorTime);
            a.rec = reinterpret_cast<const float4*>(h.rec);
            a.depth_filt = h.depth_filt; a.mask = h.mask; a.t_inv = mat4_from(h.t_inv); a.cam = h.cam; a.cols = h.cols; a.rows = h.rows; a.time = h.time;
            a.confThreshold = h.confThreshold; a.outlierCoeff = h.outlierCoeff; a.timeDelta = h.timeDelta; a.maskID = h.maskID;
            a.surfels = reinterpret_cast<const float4*>(p.surfels); a.count = p.count; a.fresh = reinterpret_cast<const float4*>(p.fresh);
            a.n_fresh = p.n_fresh; a.total_bound = p.total_bound; a.staged = reinterpret_cast<float4*>(p.staged); a.flags = p.flags;
            // (the XCD-ordered runs need a share that is a multiple of 8 x 64 workgroups: for an object model of a few thousand surfels
            // that padding would be a third again of its workgroups, all of them empty)
            // (in workgroups of kCleanB threads: the runs cover the same surfels as `chunk` workgroups of kB did)
            const int cchunk = chunk * (kB / kCleanB);
            const long long cgrid = (4ll * p.total_bound + kCleanB - 1) / kCleanB;
            a.xcd = cgrid >= 4ll * 8 * cchunk ? cchunk : 0;
#ifdef CF_ABLATE
            static const int clean_abl = getenv("CF_CLEAN_ABLATE") ? atoi(getenv("CF_CLEAN_ABLATE")) : 0;
            a.abl = clean_abl;
#endif
            { const int q = 8 * (a.xcd > 1 ? a.xcd : 1); blocks[k] = p.total_bound > 0 ? (int)((cgrid + q - 1) / q * q) : 0; }
            any += blocks[k];
        }
        if (any) clean_kernel<<<batch_layout(B, blocks, nb), kCleanB, 0, s>>>(B);
    }
}
void launch_clean(hipStream_t s, const float* surfels, const unsigned* count, const float* fresh, const unsigned* n_fresh, unsigned total_bound,
                  const SurfelCleanArgs& h, float* staged, unsigned* flags)
{
    const CleanPassArgs a{h, surfels, count, fresh, n_fresh, total_bound, staged, flags};
    launch_clean_batch(s, &a, 1);
}
void launch_set_count(hipStream_t s, unsigned* out, unsigned v) { set_count_kernel<<<1, 1, 0, s>>>(out, v); }

}  // namespace cf
