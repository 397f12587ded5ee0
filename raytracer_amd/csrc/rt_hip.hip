// rt_hip.hip -- HIP kernels and the C ABI (include/rt_hip.h) of the MI355X
// renderer backend.  gfx950 only; build with -ffp-contract=off (see rt_core.h).
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstring>
#include <type_traits>
#include <mutex>
#include <new>
#include <string>
#include <vector>

#include "../../include/rt_hip.h"
#include "rt_core.h"
#include "rt_lane.h"
#include "rt_pixel.h"
#include "rt_scene.h"

namespace {

thread_local std::string g_last_error;

int fail(int code, const std::string &msg)
{
    g_last_error = msg;
    return code;
}

#define RT_HIP_CHECK(expr)                                                                       \
    do {                                                                                         \
        hipError_t e_ = (expr);                                                                  \
        if (e_ != hipSuccess)                                                                    \
            return fail(RT_ERR_DEVICE, std::string(#expr) + ": " + hipGetErrorString(e_));       \
    } while (0)

// ---------------------------------------------------------------------------
// Workspace of rt_render_frames_noalloc: a fixed device array, so that entry point never allocates.
// Layout: [0, kScratchHeader) the work counters of the launch in flight; then the rgba16float
// colours of the frames of one launch (what rt_render_frames keeps in a hipMalloc'ed batch buffer).
// Every GPU the library runs on holds its own copy of the array until the process ends.  In host
// code the array's name is only a handle: its device address comes from hipGetSymbolAddress, once per
// device (scratch_of), and only that pointer is ever passed to kernels and memsets.
// ---------------------------------------------------------------------------
constexpr uint32_t kQueues = 8;       // work queues of a launch of the persistent kernel: one per XCD
constexpr uint32_t kQueueStride = 16; // their counters are 64 bytes apart
constexpr uint32_t kLaunchCounters = kQueues * kQueueStride; // counters one launch uses (128 dwords)
constexpr size_t kScratchHeader = (kLaunchCounters * sizeof(unsigned int) + 255) / 256 * 256;
static_assert(kScratchHeader >= kLaunchCounters * sizeof(unsigned int) && kScratchHeader % 256 == 0,
              "the counters of a launch must fit in front of the colours");
constexpr size_t kScratchColorBytes = (size_t)64 << 20; // 8,388,608 pixel-frames: one 3840x2160 frame, four 1920x1080 frames
static __device__ __align__(256) unsigned char scratch[kScratchHeader + kScratchColorBytes];

constexpr int kMaxDevices = 64;
void *g_scratch_ptr[kMaxDevices] = {nullptr};
std::mutex g_scratch_mutex; // contexts may be created and used from one host thread per device

// Device address of `scratch` on `device`, which must be the current device.  Resolved on the first
// rt_render_frames_noalloc call on that device; the other entry points never need it.
int scratch_of(int device, void **out)
{
    if (device < 0 || device >= kMaxDevices)
        return fail(RT_ERR_UNSUPPORTED, "rt_render_frames_noalloc: device index beyond the library's table of 64 devices");
    std::lock_guard<std::mutex> lock(g_scratch_mutex);
    if (!g_scratch_ptr[device]) {
        void *p = nullptr;
        RT_HIP_CHECK(hipGetSymbolAddress(&p, HIP_SYMBOL(scratch)));
        if (!p) return fail(RT_ERR_DEVICE, "hipGetSymbolAddress returned a null workspace address");
        g_scratch_ptr[device] = p;
    }
    *out = g_scratch_ptr[device];
    return RT_OK;
}

// Which image rows a context renders: local row l maps to global row
// ((l / rows) * count + index) * rows + l % rows  (8-row stripes dealt round-robin).
struct StripeMap {
    int rows, index, count;
};

__host__ __device__ inline int stripe_global_row(const StripeMap &m, int local)
{
    return ((local / m.rows) * m.count + m.index) * m.rows + local % m.rows;
}

// ---------------------------------------------------------------------------
// Per-lane traversal stack in LDS: entry k of lane l lives at base[k * BLOCK + l],
// so one wave's pushes/pops touch 64 consecutive dwords (no bank conflicts).
// ---------------------------------------------------------------------------
#ifndef RT_XCC_ID_QUEUE
#define RT_XCC_ID_QUEUE 0 // 1: a wave's first work queue is chosen by the XCC_ID hardware register instead of blockIdx % 8
#endif
#ifndef RT_PACKED_LINKS
#define RT_PACKED_LINKS 1 // small scenes: child links read from the low bytes of the node planes, no link vector fetch
#endif
#ifndef RT_PRIO
// 1: a wave runs its traversal turns at raised priority (s_setprio 2) and everything else at 0, so that on a SIMD
// the waves that are about to issue node and triangle fetches go before the ones doing the long arithmetic of a
// shading pass.  Measured against 0: suzanne 6.82 -> 6.75 ms/frame, boxes 8.26 -> 8.17, 3840x2160 27.4 -> 27.1, 991k
// triangles 10.1 -> 9.95 (mesh seen 11.97 -> 11.8), height field 16.1 -> 15.9, spheres 640x360 and 1 spp unchanged.
// The other way round (shading raised) loses 0.5-1 %; priority 3, priority for the node step only, or 2 / 1 / 0 for
// traversal / shading / the rest gain less.
#define RT_PRIO 1
#endif
#ifndef RT_TOUCH_TOP
#define RT_TOUCH_TOP 1
#endif
#ifndef RT_LEAF_LOOP
#define RT_LEAF_LOOP 0 // small scenes; 1: a triangle step tests every triangle of the leaf, not only the next one
#endif
#ifndef RT_LARGE_LEAF_LOOP
#define RT_LARGE_LEAF_LOOP 1 // the same for large scenes, whose leaves are fuller
#endif
#ifndef RT_LDS_PARK
#define RT_LDS_PARK 2
#endif
#ifndef RT_PARK2_STACK
#define RT_PARK2_STACK 22
#endif
#ifndef RT_PARK2_SLOTS
#define RT_PARK2_SLOTS 15
#endif
#ifndef RT_LARGE_STACK
#define RT_LARGE_STACK 12 // large scenes (32-bit links): stack entries in LDS, the rest in global memory
#endif
#ifndef RT_LARGE_SLOTS
#define RT_LARGE_SLOTS 14
#endif
#ifndef RT_LARGE_MINW
#define RT_LARGE_MINW 6
#endif
#ifndef RT_PARK2_MINW
#define RT_PARK2_MINW 6
#endif

template <int BLOCK, class Entry, int CAP = 1 << 30, bool SPILL = false, bool QUANT = false, bool SPHERES = true>
struct LdsStack {
    static constexpr bool kQuantNodes = QUANT; // the links on this stack count 64-byte nodes (rt::lane_visit4q)
    static constexpr bool kSphereLeaves = SPHERES; // leaves may hold spheres (rt::lane_step_triangle)
    Entry *base;
    int sp;
    // SPILL: entries CAP and above live in global memory, entry j of a lane at spill[(j - CAP) * spill_stride]
    int32_t *spill;
    uint32_t spill_stride;
    __device__ void reset() { sp = 0; }
    __device__ void push(int32_t v)
    {
        if (!SPILL || sp < CAP) base[sp * BLOCK] = (Entry)v;
        else spill[(size_t)(sp - CAP) * spill_stride] = v;
        sp++;
    }
    __device__ int32_t pop()
    {
        sp--;
        if (!SPILL || sp < CAP) return (int32_t)base[sp * BLOCK];
        return spill[(size_t)(sp - CAP) * spill_stride];
    }
    __device__ bool empty() const { return sp == 0; }
    // Large scenes (32-bit entries): a triangle step that ends a leaf also touches the record that the
    // pop after it leads to, so that its cache misses overlap with the triangle test (-2.7 % on the
    // 131k-triangle height field, -2.9 % on 991k triangles).
    static constexpr bool kTouchesTop = RT_TOUCH_TOP && sizeof(Entry) == 4;
    __device__ bool top_in_lds() const { return sp > 0 && (!SPILL || sp <= CAP); }
    __device__ int32_t top() const { return (int32_t)base[(sp - 1) * BLOCK]; }
};

// How an instantiation of the persistent kernel keeps its per-lane state.
//   STACK  traversal stack entries in LDS
//   PARK   dwords of rarely touched lane state kept in LDS (0: all in registers; 14: what is set
//          aside during shadow rays; more: per-pixel values too)
//   MINW   waves per SIMD it is compiled for: 6 (80 VGPRs) needs STACK * entry size + PARK * 4 <=
//          104 bytes per lane (26 KB per block, six blocks in a CU's 160 KB); 5 (96 VGPRs) needs it
//          below 128 (at exactly 32 KB only four blocks fit)
//   LINK16 links fit 16 bits (scenes of <= 4096 triangles): 2-byte stack entries
//   SPILL  stack entries beyond STACK go to global memory (deep trees of large scenes)
//   PLANE_LINKS  16: a node's child links are read from the low bytes of its x and y planes, where the
//          builder also writes them when they fit 16 bits (BuiltScene::links_packed), instead of from
//          its link vector: six loads per node visit instead of seven (-1.4 % on suzanne, -2.1 % on
//          boxes; the same with 24-bit links for large scenes measured no gain).  Follows LINK16.
//          64: the lanes visit the 64-byte nodes (8-bit child boxes relative to the node's own; large scenes).
//   SL     the sphere-light extension (rt::lane_shade_closest<true>): seven more parked dwords per lane, in LDS
//   SPHERES  leaves of the tree may hold spheres (scenes of many spheres).  A variant of its own: the sphere test inside
//          the triangle step costs scenes without such leaves a quarter of their speed (131k-triangle height field
//          15.9 -> 21.1 ms/frame, 991k triangles 12.5 -> 16.4) through the registers it takes.
template <int STACK_, int PARK_, int MINW_, bool LINK16_, bool SPILL_, int PLANE_LINKS_ = (LINK16_ && RT_PACKED_LINKS) ? 16 : 0, bool SL_ = false,
          bool SPHERES_ = false>
struct LaneCfg {
    static constexpr int STACK = STACK_, PARK = PARK_, MINW = MINW_, PLANE_LINKS = PLANE_LINKS_;
    static constexpr bool LINK16 = LINK16_, SPILL = SPILL_, SL = SL_, SPHERES = SPHERES_;
    static constexpr int PARK_LDS = PARK_ + (SL_ ? rt::kParkSecondShadowFloats : 0); // floats per lane in LDS
};
constexpr int kSpillEntries = 64 - RT_LARGE_STACK; // trees whose traversal needs up to 64 entries

// rt::RegPark's interface with the first NLDS slots in LDS (slot k of lane l at
// base[k * BLOCK + l]) and the rest in registers.  Slot numbers are compile-time constants at
// every use, so each access resolves to one or the other.
// SECOND: the slots of the sphere-light extension's second shadow ray (rt::kParkS2D ...) are in LDS too, after
// the first NLDS.
template <int BLOCK, int NLDS, bool SECOND = false>
struct LdsPark {
    float *base;
    float regs[NLDS < rt::kParkFloats ? rt::kParkFloats - NLDS : 1];
    // position in LDS of a slot, -1 if it lives in a register
    static __device__ constexpr int lds_index(int slot)
    {
        return slot < NLDS ? slot
                           : (SECOND && slot >= rt::kParkS2D && slot < rt::kParkS2D + rt::kParkSecondShadowFloats) ? NLDS + slot - rt::kParkS2D : -1;
    }
    __device__ void put(int slot, float x)
    {
        if (lds_index(slot) >= 0) base[lds_index(slot) * BLOCK] = x;
        else regs[slot - NLDS] = x;
    }
    __device__ float get(int slot) const { return lds_index(slot) >= 0 ? base[lds_index(slot) * BLOCK] : regs[slot - NLDS]; }
};

// ---------------------------------------------------------------------------
// Lane migration between the waves of a block ("tail compaction").
//
// When the work queues of a launch have run out, a wave keeps running until the last of its lanes has
// finished its pixel, at full cost per instruction however few lanes are left.  So a wave that is down
// to a few live lanes hands them over: it writes their registers into the LDS columns of its own dead
// lanes, publishes which lanes are up for adoption and exits; waves of the same block with dead lanes
// adopt them, a lane taking over the LDS column (traversal stack, parked state) of the lane it adopts
// in place.  A path's state never changes, only the lane and wave that execute it, so results are
// unaffected.  The header below lives in LDS, one per block.
//   alive        waves that have neither exited nor committed to donating; the last one never donates
//                and leaves only once nothing is left to adopt
//   in_progress  donors between their commitment and the publication of their lanes
//   dead[w]      lanes without work in wave w, as last published by it (receiving capacity, a hint)
//   orig[w]      lanes of donor wave w that were live when it donated;  avail[w]  those not adopted yet
// Lanes of one wave never pass values to each other through LDS here: the compiler is free to run the
// two sides of "if (dead) write; if (live) read" in either order.
// ---------------------------------------------------------------------------
#ifndef RT_MIGRATE
#define RT_MIGRATE 0 // measured: shortens a single-frame launch by 3 % and costs 3 % in long launches (five more spilled VGPRs)
#endif
struct MigrateHeader {
    unsigned int alive, in_progress;
    unsigned int dead[4];
    unsigned long long orig[4], avail[4];
};

__device__ inline unsigned long long lanes_below(int lane) { return (1ull << lane) - 1ull; }

// Position of the r-th (0-based) set bit of m.
__device__ inline int nth_set_bit(unsigned long long m, int r)
{
    for (int k = 0; k < r; k++) m &= m - 1ull;
    return __builtin_ctzll(m);
}

__device__ inline uint32_t wave_sum(uint32_t v)
{
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
    return v;
}

struct FrameBuffers {
    float *accum;        // 3 planes of width*height f32
    uint16_t *frame_f16; // width*height*4 half bits
    int32_t *ids;        // width*height
    unsigned long long *counters;
    unsigned long long *wave_trace; // instrumented launches only (may be null): 4 values per wave of the grid, see rt_debug_wave_trace
};

// rgba16float store + accumulator.wgsl:44-68 for one pixel.
// With `sum` the planes collect the plain f32 sum of the frames' rgba16float colours instead
// (frame-split rendering: the division by the frame count happens after the ranks' sums are added).
__device__ inline void store_pixel(const rt::FrameParams &fp, const FrameBuffers &fb, int px, int py, rt::f3 color,
                                   int32_t primary_id, bool sum)
{
    size_t p = (size_t)py * (size_t)fp.width + (size_t)px;
    size_t plane = (size_t)fp.width * (size_t)fp.height;
    uint16_t hr = rt::f32_to_f16_bits(color.x), hg = rt::f32_to_f16_bits(color.y), hb = rt::f32_to_f16_bits(color.z);
    uint2 packed;
    packed.x = (uint32_t)hr | ((uint32_t)hg << 16);
    packed.y = (uint32_t)hb | (0x3c00u << 16);
    reinterpret_cast<uint2 *>(fb.frame_f16)[p] = packed;
    fb.ids[p] = primary_id;
    float r = rt::f16_bits_to_f32(hr), g = rt::f16_bits_to_f32(hg), b = rt::f16_bits_to_f32(hb);
    if (sum) {
        fb.accum[p] += r;
        fb.accum[plane + p] += g;
        fb.accum[2 * plane + p] += b;
    } else {
        fb.accum[p] = rt::accumulate_channel(fb.accum[p], r, fp.frame_f);
        fb.accum[plane + p] = rt::accumulate_channel(fb.accum[plane + p], g, fp.frame_f);
        fb.accum[2 * plane + p] = rt::accumulate_channel(fb.accum[2 * plane + p], b, fp.frame_f);
    }
}

// ---------------------------------------------------------------------------
// Kernel A: one thread per pixel, the shader's own loop nest
// (raytracer_compute.wgsl:280-341), one wave per 8x8 pixel tile as the
// reference's @workgroup_size(8, 8); four tiles per 256-thread block.
// ---------------------------------------------------------------------------
constexpr int kBlock = 256;
#ifndef RT_MINW
#define RT_MINW 5 // waves per SIMD the persistent kernel is compiled for when lane state stays in registers (<= 96 VGPRs)
#endif

template <bool BRUTE, int STACK, bool COUNT>
__global__ __launch_bounds__(kBlock) void render_tiles_kernel(rt::SceneView sc, rt::FrameParams fp, StripeMap sm,
                                                               int local_rows, FrameBuffers fb, bool sum)
{
    __shared__ int32_t lds_stack[BRUTE ? 1 : STACK * kBlock];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int tile_x = blockIdx.x * 4 + wave;
    const int px = tile_x * 8 + (lane & 7);
    const int ly = blockIdx.y * 8 + (lane >> 3);
    uint32_t n_rays = 0;
    rt::TestCounts cnt = {0, 0, 0, 0};
    if (px < fp.width && ly < local_rows) {
        const int py = stripe_global_row(sm, ly);
        if (py < fp.height) {
            LdsStack<kBlock, int32_t> stack;
            stack.base = lds_stack + threadIdx.x;
            stack.sp = 0;
            stack.spill = nullptr;
            stack.spill_stride = 0;
            int32_t pid;
            rt::f3 color = rt::render_pixel<BRUTE>(sc, fp, px, py, stack, n_rays, pid, COUNT ? &cnt : nullptr);
            store_pixel(fp, fb, px, py, color, pid, sum);
        }
    }
    uint32_t total = wave_sum(n_rays);
    if (lane == 0 && total) {
        atomicAdd(fb.counters, (unsigned long long)total);
        atomicAdd(fb.counters + 13, (unsigned long long)total);
    }
    if (COUNT) {
        uint32_t b = wave_sum(cnt.boxes), t = wave_sum(cnt.tris), s = wave_sum(cnt.spheres), h = wave_sum(cnt.bounces);
        if (lane == 0) {
            atomicAdd(fb.counters + 1, (unsigned long long)b);
            atomicAdd(fb.counters + 2, (unsigned long long)t);
            atomicAdd(fb.counters + 3, (unsigned long long)s);
            atomicAdd(fb.counters + 4, (unsigned long long)h);
        }
    }
}

// ---------------------------------------------------------------------------
// Kernel B: persistent waves.  Every lane runs the path tracer as the state
// machine of rt_lane.h and always owns one ray (closest-hit or shadow), so the
// lanes of a wave need not be at the same sample, bounce or ray kind:
//   - lanes that finish a pixel get the next one from a global counter of
//     (frame, pixel) units: one atomic per wave for all its idle lanes, ranks
//     by ballot + prefix count; units are numbered 8x8-tile-major;
//   - traversal steps run while enough lanes are still traversing; when the
//     share of traversing lanes drops below a threshold, the lanes whose ray is
//     finished are shaded and given their next ray, then traversal resumes
//     (suspended lanes keep their traversal stack in LDS);
//   - triangle tests and the hand-over after a finished shadow ray are postponed
//     until enough lanes wait for them.
// Units are (frame, pixel) pairs of a batch of frames, frame-major.  The frame
// colours go to a batch buffer (or, for a single frame, a staging slot) and
// accumulate_batch_kernel applies the recurrence in frame order; FUSED instead
// accumulates a finished pixel in place (single frames with
// RT_TUNE_OVERLAP_FRAMES 0).
// ---------------------------------------------------------------------------
struct SchedParams {
    int shade_num;    // shade when traversing lanes * 8 < active lanes * shade_num
    int leaf_lanes;   // visit leaves when at least this many lanes wait at one (or no lane is at an inner node)
    int sdone_lanes;  // finish shadow rays when at least this many lanes wait for it (or no lane is traversing)
    int refill_lanes; // fetch new pixels when at least this many lanes are idle (or the whole wave is)
    int shade_wait;   // shade anyway once finished lanes have waited this many traversal turns (0: never)
    int migrate_lanes; // once the queues are empty, a wave down to this many live lanes hands them to other waves of its block (0: never)
};

struct BatchParams {
    uint32_t first_frame; // frameIndex of batch frame 0
    uint32_t n_frames;
    uint2 *colors;        // [n_frames][local_rows * width] rgba16float bit patterns (unused when FUSED)
    uint32_t stride;      // frameIndex step between batch frames (1, or the rank count when frames are dealt to ranks)
    bool sum;             // accumulator planes collect plain sums of the frame colours
    int32_t *ids;         // [height * width] where the last batch frame's primary-hit ids go: the context's id
                          // plane, or a staging plane that the accumulate kernel copies from (overlapped frames)
    bool copy_ids;        // accumulate kernel: copy `ids` into the context's id plane
    // Hand-out order of the tiles (see order_tiles_kernel): position p of queue q is tile tile_order[q_first + p]
    // (identity when null); tile_cost, when not null, collects per tile the cost of its most expensive unit in this launch.
    const uint32_t *tile_order;
    unsigned int *tile_cost;
};

// The LDS column of a lane (its park slots, then its stack entries) addressed as dwords, and the
// register image of a path that is handed from one lane to another through such columns.
template <class CFG>
struct LaneColumn {
    using Entry = typename std::conditional<CFG::LINK16, int16_t, int32_t>::type;
    static constexpr int kParkRegs = CFG::PARK < rt::kParkFloats ? rt::kParkFloats - CFG::PARK : 0;
    static constexpr int kDwords = CFG::PARK_LDS + CFG::STACK * (int)sizeof(Entry) / 4;
    static constexpr int kImage = 19 + kParkRegs;                      // dwords of register state per path
    static constexpr int kColumnsPerImage = (kImage + kDwords - 1) / kDwords;
    static constexpr int kMaxDonate = 64 / (1 + kColumnsPerImage);     // live lanes a donor can have: each needs that many dead ones
    static_assert(sizeof(Entry) == 4 || CFG::STACK % 2 == 0, "16-bit stacks must fill whole dwords");

    __device__ static uint32_t load(const Entry *st, const float *pk, int col, int i)
    {
        if (i < CFG::PARK_LDS) return rt::f2u(pk[i * kBlock + col]);
        int j = i - CFG::PARK_LDS;
        if (sizeof(Entry) == 4) return (uint32_t)st[j * kBlock + col];
        return (uint32_t)(uint16_t)st[(2 * j) * kBlock + col] | ((uint32_t)(uint16_t)st[(2 * j + 1) * kBlock + col] << 16);
    }
    __device__ static void store(Entry *st, float *pk, int col, int i, uint32_t v)
    {
        if (i < CFG::PARK_LDS) { pk[i * kBlock + col] = rt::u2f(v); return; }
        int j = i - CFG::PARK_LDS;
        if (sizeof(Entry) == 4) { st[j * kBlock + col] = (Entry)v; return; }
        st[(2 * j) * kBlock + col] = (Entry)(v & 0xffffu);
        st[(2 * j + 1) * kBlock + col] = (Entry)(v >> 16);
    }
};

__device__ inline unsigned long long uniform64(unsigned long long v)
{
    uint32_t lo = (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)v);
    uint32_t hi = (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)(v >> 32));
    return ((unsigned long long)hi << 32) | lo;
}

template <class CFG, bool COUNT, bool FUSED>
__global__ __launch_bounds__(kBlock, CFG::MINW) void render_persistent_kernel(rt::SceneView sc, rt::FrameParams fp, StripeMap sm,
                                                                    int local_rows, FrameBuffers fb, BatchParams bp,
                                                                    unsigned int *work_counter, SchedParams sched,
                                                                    int32_t *spill)
{
    constexpr int STACK = CFG::STACK, PARK = CFG::PARK;
    using StackEntry = typename std::conditional<CFG::LINK16, int16_t, int32_t>::type;
    __shared__ StackEntry lds_stack[STACK * kBlock];
    __shared__ float lds_park[CFG::PARK_LDS > 0 ? CFG::PARK_LDS * kBlock : 1];
    const int lane = threadIdx.x & 63;
    LdsStack<kBlock, StackEntry, STACK, CFG::SPILL, CFG::PLANE_LINKS == 64, CFG::SPHERES> stack;
    stack.base = lds_stack + threadIdx.x;
    stack.sp = 0;
    stack.spill = CFG::SPILL ? spill + ((size_t)blockIdx.x * kBlock + threadIdx.x) : nullptr;
    stack.spill_stride = gridDim.x * kBlock;
    LdsPark<kBlock, PARK, CFG::SL> park;
    park.base = lds_park + threadIdx.x;
#if RT_MIGRATE
    using Column = LaneColumn<CFG>;
    __shared__ MigrateHeader mig;
    const int wv = threadIdx.x >> 6;
    int col = threadIdx.x; // the LDS column (and overflow column) of the path this lane runs; changes when it adopts one
    if (threadIdx.x == 0) {
        mig.alive = kBlock / 64;
        mig.in_progress = 0u;
        for (int w = 0; w < 4; w++) { mig.dead[w] = 0u; mig.orig[w] = 0ull; mig.avail[w] = 0ull; }
    }
    __syncthreads();
#endif

    const uint32_t tiles_x = (uint32_t)(fp.width + 7) / 8u, tiles_y = (uint32_t)(local_rows + 7) / 8u;
    const uint32_t tiles_per_frame = tiles_x * tiles_y;
    // Work queues.  The (frame, pixel) units of a launch are split into kQueues queues, each with its
    // own counter: queue q holds the q-th contiguous share of a frame's tiles (a band of the image),
    // for all frames of the launch, frame-major.  A wave starts on the queue of its XCD (workgroups
    // are dealt to the eight XCDs round-robin, so that is blockIdx % 8) and moves on to the next queue
    // when its own runs out, until all are empty.  Each XCD has its own L2: as long as the queues
    // last, the camera and shadow rays of an XCD stay within one band of the image, i.e. within one
    // part of the scene, instead of every L2 holding the geometry behind the whole image.
#if RT_XCC_ID_QUEUE
    // the XCD the wave runs on, from the hardware register (HW_REG_XCC_ID = 20, bits 0..3)
    uint32_t my_queue = (uint32_t)__builtin_amdgcn_s_getreg(20 | (3 << 11)) % kQueues;
#else
    uint32_t my_queue = blockIdx.x % kQueues; // wave-uniform
#endif
    uint32_t queues_left = 0;                  // queues this wave has found empty and left behind

    rt::Lane L;
    L.mode = rt::kLaneNeedUnit;
    rt::RayCounts n_rays = {0u, 0u};
    rt::TestCounts cnt = {0, 0, 0, 0};
    rt::TestCounts *cp = COUNT ? &cnt : nullptr;

    uint32_t ph_inner_it = 0, ph_inner_ln = 0, ph_leaf_it = 0, ph_leaf_ln = 0, ph_shade_it = 0, ph_shade_ln = 0,
             ph_sdone_it = 0, ph_sdone_ln = 0; // COUNT only
    uint32_t occ_it = 0, occ_inner = 0, occ_leaf = 0, occ_sdone = 0, occ_ready = 0, occ_idle = 0, occ_dead = 0; // COUNT only: lanes per state, summed over traversal turns
    unsigned long long tc_inner = 0, tc_tri = 0, tc_sdone = 0, tc_shade = 0, tc_total = 0, tc0 = 0, tc_begin = 0; // COUNT only: wave cycles per phase
    if (COUNT) tc_begin = clock64();
    unsigned long long tr_begin = 0, tr_exhausted = 0, tr_live_at_exhaustion = 0; // COUNT only: wall-clock ticks (100 MHz)
    if (COUNT) tr_begin = wall_clock64();
    bool exhausted = false; // wave-uniform: the global counter has run out

    for (;;) {
        // ---- store finished pixels -----------------------------------------------------------
        if (L.mode == rt::kLanePixelDone) {
            rt::f3 color = rt::lane_pixel_color(fp, park);
            uint32_t tile_all = L.unit >> 6, slot = L.unit & 63u;
            uint32_t frame_k = tile_all / tiles_per_frame, tile = tile_all % tiles_per_frame;
            int px = (int)((tile % tiles_x) * 8u + (slot & 7u));
            int ly = (int)((tile / tiles_x) * 8u + (slot >> 3));
            int py = stripe_global_row(sm, ly);
            // The cost of a tile, for the order in which the next launch hands tiles out: the most rays any of its
            // units took (a launch ends when its last unit does, so a tile with one long pixel must go early however
            // cheap the rest of it is).  Units of the launch's first frame only: one frame's costs are as good as all,
            // and with every unit reporting, thousands of lanes would hit the few tiles in hand at any moment.
            if (bp.tile_cost && frame_k == 0u) atomicMax(bp.tile_cost + tile, rt::lane_unit_cost(L) + 1u);
            if (FUSED) store_pixel(fp, fb, px, py, color, rt::lane_primary_id(sc, park), bp.sum);
            else {
                size_t lp = (size_t)ly * fp.width + px;
                uint2 packed;
                packed.x = (uint32_t)rt::f32_to_f16_bits(color.x) | ((uint32_t)rt::f32_to_f16_bits(color.y) << 16);
                packed.y = (uint32_t)rt::f32_to_f16_bits(color.z) | (0x3c00u << 16);
                bp.colors[(size_t)frame_k * ((size_t)local_rows * fp.width) + lp] = packed;
                if (frame_k + 1u == bp.n_frames) bp.ids[(size_t)py * fp.width + px] = rt::lane_primary_id(sc, park);
            }
            L.mode = rt::kLaneNeedUnit;
        }

        // ---- hand out pixels to idle lanes ------------------------------------------------
        // Units are numbered tile-major (64 consecutive units = one 8x8 tile), so lanes
        // that refill together get neighbouring pixels.  A wave takes exactly as many
        // units as it has idle lanes (one atomic for all of them), once enough lanes
        // are idle or nothing else is left to do.
        // (No loop here: lanes that drew a padding pixel of an edge tile stay idle and ask again on
        // the next turn of the outer loop, which comes at once when no lane has anything to trace.)
        {
            unsigned long long need = __ballot(L.mode == rt::kLaneNeedUnit);
            uint32_t n_need = (uint32_t)__popcll(need);
            bool any_busy = __ballot(L.mode != rt::kLaneNeedUnit && L.mode != rt::kLaneDead) != 0ull;
            if (need != 0ull && exhausted) {
                if (L.mode == rt::kLaneNeedUnit) L.mode = rt::kLaneDead;
            } else if (need != 0ull && (!any_busy || n_need >= (uint32_t)sched.refill_lanes)) {
                // this queue's share of the tiles of a frame, and its length in units
                const uint32_t q_first = (uint32_t)((uint64_t)tiles_per_frame * my_queue / kQueues);
                const uint32_t q_tiles = (uint32_t)((uint64_t)tiles_per_frame * (my_queue + 1u) / kQueues) - q_first;
                const uint32_t q_units = q_tiles * bp.n_frames * 64u;
                unsigned int base = 0;
                if (lane == 0) base = atomicAdd(work_counter + my_queue * kQueueStride, n_need);
                base = (unsigned int)__builtin_amdgcn_readfirstlane((int)base);
                uint32_t rank = (uint32_t)__popcll(need & ((1ull << lane) - 1ull));
                uint32_t n = base + rank; // position in the queue
                if (L.mode == rt::kLaneNeedUnit && n < q_units) {
                    // positions count tile-major: all frames of the queue's first tile, then of its second, ...
                    uint32_t group = n >> 6, slot = n & 63u;
                    uint32_t tile_pos = group / bp.n_frames, frame_k = group % bp.n_frames;
                    uint32_t tile = q_first + tile_pos;
                    if (bp.tile_order) tile = bp.tile_order[tile];
                    int px = (int)((tile % tiles_x) * 8u + (slot & 7u));
                    int ly = (int)((tile / tiles_x) * 8u + (slot >> 3));
                    int py = stripe_global_row(sm, ly);
                    if (px < fp.width && ly < local_rows && py < fp.height) {
                        L.unit = (frame_k * tiles_per_frame + tile) * 64u + slot;
                        if (fp.max_bounces == 0u) {
                            rt::lane_empty_pixel(L, park);
                        } else {
                            rt::FrameParams f = fp;
                            if (!FUSED) f.seed_add = (uint32_t)(float)(bp.first_frame + frame_k * bp.stride) * 12345u;
                            rt::lane_begin_pixel(sc, f, L, px, py, stack, park, n_rays, cp);
                        }
                    }
                }
                // The queue ran out with this grab (lanes beyond its end got nothing and ask again): on to
                // the next one; after leaving all of them behind there is no work left.
                if (base + n_need >= q_units) {
                    my_queue = (my_queue + 1u) % kQueues;
                    if (++queues_left == kQueues) {
                        exhausted = true;
                        if (COUNT) { tr_exhausted = wall_clock64(); tr_live_at_exhaustion = (unsigned long long)__popcll(__ballot(L.mode != rt::kLaneNeedUnit && L.mode != rt::kLaneDead)); }
                    }
                }
            }
        }
#if RT_MIGRATE
        if (sched.migrate_lanes > 0) {
            if (exhausted) {
                // ---- adopt lanes that other waves of the block have put up ------------------------------
                for (int w = 0; w < 4; w++) {
                    unsigned long long dead_mask = __ballot(L.mode == rt::kLaneDead);
                    if (dead_mask == 0ull) break;
                    unsigned long long av = uniform64(__hip_atomic_load(&mig.avail[w], __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_WORKGROUP));
                    if (av == 0ull) continue;
                    int n_take = min(__popcll(av), __popcll(dead_mask));
                    unsigned long long take = 0ull, rest = av;
                    for (int k = 0; k < n_take; k++) { unsigned long long b = rest & (~rest + 1ull); take |= b; rest ^= b; }
                    unsigned long long old = 0ull;
                    if (lane == 0) old = atomicCAS(&mig.avail[w], av, av & ~take);
                    old = uniform64(old);
                    if (old != av) { w--; continue; } // another wave took some of them first: look again
                    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
                    int r = __popcll(dead_mask & lanes_below(lane));
                    if (L.mode == rt::kLaneDead && r < n_take) {
                        int from = nth_set_bit(take, r);
                        int k = __popcll(mig.orig[w] & lanes_below(from)); // rank among the donor's live lanes
                        // its registers are in the natural columns of the donor's dead lanes, rank for rank
                        unsigned long long donor_dead = ~mig.orig[w];
                        int src[Column::kColumnsPerImage];
#pragma unroll
                        for (int c = 0; c < Column::kColumnsPerImage; c++) src[c] = w * 64 + nth_set_bit(donor_dead, k * Column::kColumnsPerImage + c);
                        uint32_t img[Column::kImage];
#pragma unroll
                        for (int i = 0; i < Column::kImage; i++) img[i] = Column::load(lds_stack, lds_park, src[i / Column::kDwords], i % Column::kDwords);
                        L.unit = img[0]; L.rng = img[1]; L.bits = img[2];
                        L.mode = (int32_t)(img[3] & 0xffu);
                        stack.sp = (int)((img[3] >> 8) & 0xffu);
                        col = (int)(img[3] >> 16);
                        L.color = rt::mk3(rt::u2f(img[4]), rt::u2f(img[5]), rt::u2f(img[6]));
                        L.light = rt::mk3(rt::u2f(img[7]), rt::u2f(img[8]), rt::u2f(img[9]));
                        L.o = rt::mk3(rt::u2f(img[10]), rt::u2f(img[11]), rt::u2f(img[12]));
                        L.d = rt::mk3(rt::u2f(img[13]), rt::u2f(img[14]), rt::u2f(img[15]));
                        L.tmax = rt::u2f(img[16]); L.best_id = (int32_t)img[17]; L.cur = (int32_t)img[18];
#pragma unroll
                        for (int i = 0; i < Column::kParkRegs; i++) park.regs[i] = rt::u2f(img[19 + i]);
                        L.rb = rt::make_raybox(L.o, L.d); // a function of the ray alone
                        stack.base = lds_stack + col;
                        park.base = lds_park + col;
                        if (CFG::SPILL) stack.spill = spill + ((size_t)blockIdx.x * kBlock + col);
                    }
                }
                unsigned long long live = __ballot(L.mode != rt::kLaneDead);
                int n_live = __popcll(live);
                if (n_live == 0) {
                    // ---- leave, unless this is the block's last wave and lanes are on offer or about to be ----
                    unsigned int before = 0u;
                    if (lane == 0) { mig.dead[wv] = 0u; before = __hip_atomic_fetch_sub(&mig.alive, 1u, __ATOMIC_ACQ_REL, __HIP_MEMORY_SCOPE_WORKGROUP); }
                    before = (unsigned int)__builtin_amdgcn_readfirstlane((int)before);
                    if (before > 1u) break;
                    while (__hip_atomic_load(&mig.in_progress, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_WORKGROUP) != 0u) __builtin_amdgcn_s_sleep(2);
                    unsigned long long offered = 0ull;
                    for (int w = 0; w < 4; w++) offered |= __hip_atomic_load(&mig.avail[w], __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_WORKGROUP);
                    if (uniform64(offered) == 0ull) break;
                    if (lane == 0) __hip_atomic_fetch_add(&mig.alive, 1u, __ATOMIC_ACQ_REL, __HIP_MEMORY_SCOPE_WORKGROUP);
                    continue; // adopt them on the next turn
                }
                if (lane == 0 && mig.dead[wv] != (unsigned int)(64 - n_live)) mig.dead[wv] = (unsigned int)(64 - n_live);
                // ---- hand over the remaining lanes when they are few and others have room for them -----
                if (n_live <= sched.migrate_lanes && n_live <= Column::kMaxDonate) {
                    unsigned int room = 0u;
                    for (int w = 0; w < 4; w++)
                        if (w != wv) room += __hip_atomic_load(&mig.dead[w], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                    room = (unsigned int)__builtin_amdgcn_readfirstlane((int)room);
                    unsigned int before = 0u;
                    if (room >= (unsigned int)n_live && lane == 0) {
                        __hip_atomic_fetch_add(&mig.in_progress, 1u, __ATOMIC_ACQ_REL, __HIP_MEMORY_SCOPE_WORKGROUP);
                        before = __hip_atomic_fetch_sub(&mig.alive, 1u, __ATOMIC_ACQ_REL, __HIP_MEMORY_SCOPE_WORKGROUP);
                        if (before <= 1u) { // the last wave stays
                            __hip_atomic_fetch_add(&mig.alive, 1u, __ATOMIC_ACQ_REL, __HIP_MEMORY_SCOPE_WORKGROUP);
                            __hip_atomic_fetch_sub(&mig.in_progress, 1u, __ATOMIC_ACQ_REL, __HIP_MEMORY_SCOPE_WORKGROUP);
                        }
                    }
                    before = (unsigned int)__builtin_amdgcn_readfirstlane((int)before);
                    if (before > 1u) {
                        // the natural columns of this wave's dead lanes are free (a lane that runs an adopted
                        // path uses the column it adopted): they take the register images, the k-th live
                        // lane's going to the k-th dead lane's column (both sides work that out from the mask
                        // of live lanes alone, nothing is passed between the lanes of this wave)
                        if (L.mode != rt::kLaneDead) {
                            int k = __popcll(live & lanes_below(lane));
                            int dst_col[Column::kColumnsPerImage];
#pragma unroll
                            for (int c = 0; c < Column::kColumnsPerImage; c++) dst_col[c] = wv * 64 + nth_set_bit(~live, k * Column::kColumnsPerImage + c);
                            uint32_t img[Column::kImage];
                            img[0] = L.unit; img[1] = L.rng; img[2] = L.bits;
                            img[3] = (uint32_t)L.mode | ((uint32_t)stack.sp << 8) | ((uint32_t)col << 16);
                            img[4] = rt::f2u(L.color.x); img[5] = rt::f2u(L.color.y); img[6] = rt::f2u(L.color.z);
                            img[7] = rt::f2u(L.light.x); img[8] = rt::f2u(L.light.y); img[9] = rt::f2u(L.light.z);
                            img[10] = rt::f2u(L.o.x); img[11] = rt::f2u(L.o.y); img[12] = rt::f2u(L.o.z);
                            img[13] = rt::f2u(L.d.x); img[14] = rt::f2u(L.d.y); img[15] = rt::f2u(L.d.z);
                            img[16] = rt::f2u(L.tmax); img[17] = (uint32_t)L.best_id; img[18] = (uint32_t)L.cur;
#pragma unroll
                            for (int i = 0; i < Column::kParkRegs; i++) img[19 + i] = rt::f2u(park.regs[i]);
#pragma unroll
                            for (int i = 0; i < Column::kImage; i++) Column::store(lds_stack, lds_park, dst_col[i / Column::kDwords], i % Column::kDwords, img[i]);
                        }
                        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup"); // images, stacks (LDS and overflow columns) before the offer
                        if (lane == 0) {
                            mig.orig[wv] = live;
                            mig.dead[wv] = 0u;
                            __hip_atomic_store(&mig.avail[wv], live, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP);
                            __hip_atomic_fetch_sub(&mig.in_progress, 1u, __ATOMIC_ACQ_REL, __HIP_MEMORY_SCOPE_WORKGROUP);
                        }
                        L.mode = rt::kLaneDead;
                        break;
                    }
                }
            }
        } else
#endif
        if (__ballot(L.mode != rt::kLaneDead) == 0ull) break;

        // ---- traversal steps ----------------------------------------------------------------
        int turns_waiting = 0; // turns of this loop during which finished lanes have been waiting to be shaded
        if (RT_PRIO) __builtin_amdgcn_s_setprio(2);
        for (;;) {
            int n_trav = __popcll(__ballot(L.mode == rt::kLaneTraverse));
            int n_sdone = __popcll(__ballot(L.mode == rt::kLaneShadowDone));
            if (n_trav + n_sdone == 0) break;
            int n_ready = __popcll(__ballot(L.mode == rt::kLaneShadeClosest || L.mode == rt::kLanePixelDone));
            if (n_ready != 0 && (n_trav + n_sdone) * 8 < (n_trav + n_sdone + n_ready) * sched.shade_num) break;
            // long traversals of a few lanes must not hold the finished ones for ever
            if (n_ready != 0 && sched.shade_wait > 0 && ++turns_waiting >= sched.shade_wait) break;

            bool inner = L.mode == rt::kLaneTraverse && rt::link_is_inner(L.cur);
            if (COUNT) {
                occ_it++;
                occ_inner += __popcll(__ballot(inner));
                occ_leaf += __popcll(__ballot(L.mode == rt::kLaneTraverse && rt::link_is_leaf(L.cur)));
                occ_sdone += n_sdone;
                occ_ready += n_ready;
                occ_idle += __popcll(__ballot(L.mode == rt::kLaneNeedUnit));
                occ_dead += __popcll(__ballot(L.mode == rt::kLaneDead));
            }
            if (COUNT) { int n_in = __popcll(__ballot(inner)); if (n_in) { ph_inner_it++; ph_inner_ln += n_in; } tc0 = clock64(); }
            if (inner) rt::lane_step_node<CFG::PLANE_LINKS>(sc, L, stack, cp);
            if (COUNT) tc_inner += clock64() - tc0;

            bool leaf = L.mode == rt::kLaneTraverse && rt::link_is_leaf(L.cur);
            int n_leaf = __popcll(__ballot(leaf));
            bool any_inner = __ballot(L.mode == rt::kLaneTraverse && rt::link_is_inner(L.cur)) != 0ull;
            if (n_leaf != 0 && (n_leaf >= sched.leaf_lanes || !any_inner)) {
                if (COUNT) { ph_leaf_it++; ph_leaf_ln += n_leaf; tc0 = clock64(); }
                if (CFG::LINK16 ? RT_LEAF_LOOP : RT_LARGE_LEAF_LOOP) {
                    // All triangles of the leaf in one go (lanes with fewer triangles idle in the later rounds).  Pays
                    // where leaves are full: a lane's ray waits one memory round trip per turn, and the later
                    // triangles of a leaf lie in the cache lines its first one brought in (+14 % on 991k triangles,
                    // +3 % on the 131k-triangle height field; suzanne's 581 leaves hold 982 triangles and lose 0.5 %).
                    int rounds = leaf ? (int)((uint32_t)~L.cur & 7u) + 1 : 0;
                    while (__ballot(rounds > 0) != 0ull) {
                        if (rounds > 0) {
                            rt::lane_step_triangle(sc, L, stack, cp);
                            rounds = L.mode == rt::kLaneTraverse ? rounds - 1 : 0;
                        }
                    }
                } else if (leaf) rt::lane_step_triangle(sc, L, stack, cp);
                if (COUNT) tc_tri += clock64() - tc0;
            }

            // shadow rays that have finished: add their light, start the parked ray
            bool sdone = L.mode == rt::kLaneShadowDone;
            n_sdone = __popcll(__ballot(sdone));
            if (n_sdone != 0 && (n_sdone >= sched.sdone_lanes || __ballot(L.mode == rt::kLaneTraverse) == 0ull)) {
                if (COUNT) { ph_sdone_it++; ph_sdone_ln += n_sdone; tc0 = clock64(); }
                if (sdone) rt::lane_shadow_done<CFG::SL>(sc, fp, L, stack, park, n_rays, cp);
                if (COUNT) tc_sdone += clock64() - tc0;
            }
        }

        // ---- shade the lanes whose closest-hit ray is finished ------------------------------------
        if (COUNT) { int n_sh = __popcll(__ballot(L.mode == rt::kLaneShadeClosest)); if (n_sh) { ph_shade_it++; ph_shade_ln += n_sh; } }
        if (COUNT) tc0 = clock64();
        if (RT_PRIO) __builtin_amdgcn_s_setprio(0);
        if (L.mode == rt::kLaneShadeClosest) rt::lane_shade_closest<CFG::SL>(sc, fp, L, stack, park, n_rays, cp);
        if (COUNT) tc_shade += clock64() - tc0;
    }
    if (COUNT) tc_total = clock64() - tc_begin;
    if (COUNT && fb.wave_trace && lane == 0) {
        unsigned long long *t = fb.wave_trace + 4 * ((size_t)blockIdx.x * (kBlock / 64) + (threadIdx.x >> 6));
        t[0] = tr_begin; t[1] = tr_exhausted; t[2] = wall_clock64(); t[3] = tr_live_at_exhaustion;
    }

    uint32_t total = wave_sum(n_rays.calls), total_trav = wave_sum(n_rays.traversed);
    if (lane == 0 && total) {
        atomicAdd(fb.counters, (unsigned long long)total);
        atomicAdd(fb.counters + 13, (unsigned long long)total_trav);
    }
    if (COUNT) {
        uint32_t b = wave_sum(cnt.boxes), t = wave_sum(cnt.tris), s = wave_sum(cnt.spheres), h = wave_sum(cnt.bounces);
        if (lane == 0) {
            atomicAdd(fb.counters + 1, (unsigned long long)b);
            atomicAdd(fb.counters + 2, (unsigned long long)t);
            atomicAdd(fb.counters + 3, (unsigned long long)s);
            atomicAdd(fb.counters + 4, (unsigned long long)h);
            atomicAdd(fb.counters + 5, (unsigned long long)ph_inner_it);
            atomicAdd(fb.counters + 6, (unsigned long long)ph_inner_ln);
            atomicAdd(fb.counters + 7, (unsigned long long)ph_leaf_it);
            atomicAdd(fb.counters + 8, (unsigned long long)ph_leaf_ln);
            atomicAdd(fb.counters + 9, (unsigned long long)ph_shade_it);
            atomicAdd(fb.counters + 10, (unsigned long long)ph_shade_ln);
            atomicAdd(fb.counters + 11, (unsigned long long)ph_sdone_it);
            atomicAdd(fb.counters + 12, (unsigned long long)ph_sdone_ln);
            atomicAdd(fb.counters + 16, tc_inner);
            atomicAdd(fb.counters + 17, tc_tri);
            atomicAdd(fb.counters + 18, tc_sdone);
            atomicAdd(fb.counters + 19, tc_shade);
            atomicAdd(fb.counters + 20, tc_total);
            atomicAdd(fb.counters + 24, (unsigned long long)occ_it);
            atomicAdd(fb.counters + 25, (unsigned long long)occ_inner);
            atomicAdd(fb.counters + 26, (unsigned long long)occ_leaf);
            atomicAdd(fb.counters + 27, (unsigned long long)occ_sdone);
            atomicAdd(fb.counters + 28, (unsigned long long)occ_ready);
            atomicAdd(fb.counters + 29, (unsigned long long)occ_idle);
            atomicAdd(fb.counters + 30, (unsigned long long)occ_dead);
        }
    }
}

// accumulator.wgsl:44-68 applied for the frames of a batch in order; also leaves
// the last frame in the rgba16float intermediate buffer.
__global__ void accumulate_batch_kernel(FrameBuffers fb, BatchParams bp, int width, int height, int local_rows, StripeMap sm)
{
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    size_t n = (size_t)local_rows * width;
    if (i >= n) return;
    int ly = (int)(i / width), x = (int)(i % width);
    int y = stripe_global_row(sm, ly);
    if (y >= height) return;
    size_t p = (size_t)y * width + x, plane = (size_t)width * height;
    float r = fb.accum[p], g = fb.accum[plane + p], b = fb.accum[2 * plane + p];
    uint2 packed = make_uint2(0u, 0u);
    for (uint32_t k = 0; k < bp.n_frames; k++) {
        packed = bp.colors[(size_t)k * n + i];
        float frame_f = (float)(bp.first_frame + k * bp.stride);
        float cr = rt::f16_bits_to_f32((uint16_t)(packed.x & 0xffffu)), cg = rt::f16_bits_to_f32((uint16_t)(packed.x >> 16)),
              cb = rt::f16_bits_to_f32((uint16_t)(packed.y & 0xffffu));
        if (bp.sum) {
            r += cr; g += cg; b += cb;
        } else {
            r = rt::accumulate_channel(r, cr, frame_f);
            g = rt::accumulate_channel(g, cg, frame_f);
            b = rt::accumulate_channel(b, cb, frame_f);
        }
    }
    fb.accum[p] = r;
    fb.accum[plane + p] = g;
    fb.accum[2 * plane + p] = b;
    reinterpret_cast<uint2 *>(fb.frame_f16)[p] = packed;
    if (bp.copy_ids) fb.ids[p] = bp.ids[p];
}

// planes := saturate(planes / divisor): closes a frame-split render (accumulator.wgsl:57-60 for
// the whole sum at once).
__global__ void finish_sums_kernel(float *planes, size_t n, float divisor)
{
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) planes[i] = fminf(fmaxf(planes[i] / divisor, 0.0f), 1.0f);
}

// Hand-out order of a launch's tiles: within each work queue, the tiles that held the longest units of the
// previous launch come first.  A launch ends when its last unit ends, and a unit is serial work (the
// samples of a pixel-frame share one random stream): whatever is handed out last should be short.
// One block per queue.  cost: per tile, the cost of its most expensive unit in the previous launch (rt::lane_unit_cost + 1;
// all zero before the first);
// cls: scratch of one entry per tile; order: out, a permutation of each queue's tiles, most expensive
// of 256 cost classes first.  Every tile's cost is read exactly once, so the result is a permutation
// whatever the values are; the order within a class is whatever the atomic counters make it, which no result
// depends on.
__global__ __launch_bounds__(256) void order_tiles_kernel(const unsigned int *cost, uint32_t *cls, uint32_t *order, uint32_t tiles_per_frame)
{
    __shared__ unsigned int top, bins[256];
    const uint32_t q = blockIdx.x;
    const uint32_t q_first = (uint32_t)((uint64_t)tiles_per_frame * q / kQueues);
    const uint32_t q_tiles = (uint32_t)((uint64_t)tiles_per_frame * (q + 1u) / kQueues) - q_first;
    if (threadIdx.x == 0) top = 0u;
    bins[threadIdx.x] = 0u;
    __syncthreads();
    unsigned int mine = 0u;
    for (uint32_t i = threadIdx.x; i < q_tiles; i += 256u) {
        unsigned int c = cost[q_first + i];
        cls[q_first + i] = c;
        mine = max(mine, c);
    }
    atomicMax(&top, mine);
    __syncthreads();
    const unsigned int most = top;
    for (uint32_t i = threadIdx.x; i < q_tiles; i += 256u) {
        unsigned int c = cls[q_first + i];
        unsigned int k = most ? 255u - (unsigned int)((uint64_t)c * 255u / most) : 0u; // 0 = most expensive
        cls[q_first + i] = k;
        atomicAdd(&bins[k], 1u);
    }
    __syncthreads();
    if (threadIdx.x == 0) { // exclusive prefix sums: where each class starts
        unsigned int run = 0u;
        for (int k = 0; k < 256; k++) { unsigned int n = bins[k]; bins[k] = run; run += n; }
    }
    __syncthreads();
    for (uint32_t i = threadIdx.x; i < q_tiles; i += 256u) {
        unsigned int at = atomicAdd(&bins[cls[q_first + i]], 1u);
        order[q_first + at] = q_first + i;
    }
}

// The canvas output of the accumulator pass (`return vec4(accumulatedColor, 1.0)`, accumulator.wgsl:67, into an
// 8-bit unorm target): RGBA8 of the accumulator planes, no gamma; value * 255 rounded to nearest.
__global__ void present_rgba8_kernel(const float *accum, uint32_t *rgba, size_t px)
{
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= px) return;
    uint32_t out = 0xff000000u;
    for (int ch = 0; ch < 3; ch++) {
        float v = accum[ch * px + i];
        double scaled = floor((double)v * 255.0 + 0.5);
        uint32_t q = scaled >= 0.0 && scaled <= 255.0 ? (uint32_t)scaled : (scaled > 255.0 ? 255u : 0u); // NaN -> 0
        out |= q << (8 * ch);
    }
    rgba[i] = out;
}

__global__ void fill_i32_kernel(int32_t *p, int32_t v, size_t n)
{
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) p[i] = v;
}

// rows of the three planes <-> a dense block of owned rows
__global__ void export_rows_kernel(const float *accum, float *dst, int width, int height, int local_rows, StripeMap sm)
{
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    size_t n = (size_t)local_rows * width;
    if (i >= n) return;
    int ly = (int)(i / width), x = (int)(i % width);
    int y = stripe_global_row(sm, ly);
    size_t plane = (size_t)width * height;
    for (int c = 0; c < 3; c++) dst[c * n + i] = accum[c * plane + (size_t)y * width + x];
}

// src_plane: elements between the planes of the block (local_rows * width when the block is dense)
__global__ void import_rows_kernel(float *accum, const float *src, int width, int height, int local_rows, StripeMap sm,
                                   size_t src_plane)
{
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    size_t n = (size_t)local_rows * width;
    if (i >= n) return;
    int ly = (int)(i / width), x = (int)(i % width);
    int y = stripe_global_row(sm, ly);
    size_t plane = (size_t)width * height;
    for (int c = 0; c < 3; c++) accum[c * plane + (size_t)y * width + x] = src[c * src_plane + i];
}

int owned_rows_of(int height, const StripeMap &m)
{
    int n = 0;
    for (int y = 0; y < height; y++)
        if ((y / m.rows) % m.count == m.index) n++;
    return n;
}

template <class T>
int upload(const std::vector<T> &v, T **dptr, hipStream_t s)
{
    *dptr = nullptr;
    size_t bytes = std::max<size_t>(v.size() * sizeof(T), 16);
    RT_HIP_CHECK(hipMalloc((void **)dptr, bytes));
    if (!v.empty()) RT_HIP_CHECK(hipMemcpyAsync(*dptr, v.data(), v.size() * sizeof(T), hipMemcpyHostToDevice, s));
    return RT_OK;
}

} // namespace

struct rt_context {
    int device = 0;
    int width = 0, height = 0;
    StripeMap sm{8, 0, 1};
    int local_rows = 0;
    hipStream_t stream = nullptr;
    hipEvent_t ev_start = nullptr, ev_stop = nullptr;
    bool timed = false;
    uint32_t timed_frames = 0; // frames between ev_start and ev_stop
    void *scratch = nullptr; // this device's copy of the fixed workspace (rt_render_frames_noalloc)
    bool no_alloc = false;   // set for the duration of an rt_render_frames_noalloc call

    FrameBuffers fb{nullptr, nullptr, nullptr, nullptr, nullptr};

    // scene
    rt::BuiltScene host_scene;
    rt::SceneView view{};
    uint32_t *d_tri_orig = nullptr;
    rt::f4 *d_tri_shade_leaf = nullptr;
    rt::f4 *d_nodes4q = nullptr, *d_sph_leaf = nullptr;
    uint32_t *d_sph_orig = nullptr;
    rt::f4 *d_nodes = nullptr, *d_nodes4 = nullptr, *d_tri_isect = nullptr, *d_tri_shade = nullptr, *d_tri_raw = nullptr, *d_spheres = nullptr,
           *d_lights = nullptr;
    bool has_scene = false;
    size_t scene_bytes = 0;

    float uniforms[20] = {0};
    bool has_uniforms = false;
    uint32_t max_bounces = 6;
    int mode = RT_MODE_BVH;
    bool counting = false;
    unsigned long long tests_taken_rays = 0; // rays already reported by rt_take_triangle_tests

    // persistent-kernel launch state
    int n_cus = 0;
    unsigned int *d_work = nullptr; // ring of work-queue counters, kLaunchCounters per launch
    uint32_t work_used = 0;
    uint2 *d_batch = nullptr;       // frame colours of a batch
    size_t batch_capacity = 0;      // in frames
    // Single-frame launches (rt_render_frame) alternate between two side streams so that the next
    // frame's trace fills the CUs the previous one leaves idle while its last pixels finish; each
    // has a staging slot for the frame colours and ids, and the accumulate step stays on `stream`,
    // in frame order.
    static constexpr int kMaxSlots = 4;
    hipStream_t aux[kMaxSlots] = {nullptr, nullptr, nullptr, nullptr};
    hipEvent_t trace_done[kMaxSlots] = {nullptr, nullptr, nullptr, nullptr}, slot_free[kMaxSlots] = {nullptr, nullptr, nullptr, nullptr};
    uint2 *d_slot[kMaxSlots] = {nullptr, nullptr, nullptr, nullptr};
    int32_t *d_slot_ids[kMaxSlots] = {nullptr, nullptr, nullptr, nullptr};
    bool slot_used[kMaxSlots] = {false, false, false, false};
    int32_t *d_spill[1 + kMaxSlots] = {nullptr, nullptr, nullptr, nullptr, nullptr}; // traversal stack overflow, per stream (main, side streams)
    size_t spill_bytes[1 + kMaxSlots] = {0, 0, 0, 0, 0};
    // Tile hand-out order, per stream (main, side streams): rays per tile of the stream's last two launches
    // (one being written, one being read), the order made from it, and scratch of the ordering kernel.
    unsigned int *d_tile_cost[1 + kMaxSlots][2] = {};
    uint32_t *d_tile_order[1 + kMaxSlots] = {}, *d_tile_cls[1 + kMaxSlots] = {};
    int tile_cost_phase[1 + kMaxSlots] = {};
    // Presented image (rt_present): two slots, each a device block, a pinned host block and the event that ends its
    // copy; the copies run on a stream of their own.
    hipStream_t copy_stream = nullptr;
    uint32_t *d_present[2] = {nullptr, nullptr};
    uint32_t *h_present[2] = {nullptr, nullptr};
    hipEvent_t present_ready[2] = {nullptr, nullptr}, present_done[2] = {nullptr, nullptr};
    bool present_used[2] = {false, false};
    uint64_t present_serial[2] = {0, 0}; // launch_serial at the time the slot's image was asked for
    uint32_t present_frame[2] = {0, 0};  // frameIndex of the last frame in that image
    int present_next = 0;
    uint64_t launch_serial = 0;          // counts the calls that changed the accumulator planes (launches, imports, clears): has the image changed since the last present?
    uint32_t launched_frame = 0;         // frameIndex of the last frame launched
    int tune_tile_order = 1;
    int tune_linear_spheres = 8;
    int next_slot = 0;
    // Submission coalescing of rt_render_frame (see flush_pending): frames pending_first ..
    // pending_first + pending_n - 1 have been asked for and not launched yet.
    uint32_t pending_first = 0, pending_n = 0;
    int tune_coalesce = 16;             // most frames held back; <= 1: every frame is launched by its own call
    hipEvent_t flush_done[2] = {nullptr, nullptr}; // ends of the last two launches made by flush_pending
    bool flush_used[2] = {false, false};
    int next_flush = 0;
    int tune_overlap = 3; // side streams (= frames in flight) of the single-frame path; 0: fused path on the main stream

    int stream_index(hipStream_t s) const // 0 main, 1.. side streams
    {
        for (int k = 0; k < kMaxSlots; k++)
            if (s == aux[k]) return 1 + k;
        return 0;
    }
    size_t wave_trace_len = 0, wave_trace_used = 0; // fb.wave_trace: values allocated / written by the last instrumented launch
    float *d_row_staging = nullptr; // device block the host-memory row exchange (rt_read_owned_rows / rt_write_rows) goes through
    size_t row_staging_bytes = 0;
    int tune_max_stack = 64;
    int tune_shade_wait = 12;
    int tune_migrate = RT_MIGRATE ? 16 : 0;
    bool debug_view = false;    // opt-in extension: uniforms.debugEnabled draws the triangles' normal markers (brute-force kernel)
    bool sphere_lights = false; // opt-in extension: emissive spheres are sampled too (one-thread-per-pixel kernel)
    int tune_shade_num = -1, tune_leaf_lanes = -1; // -1: by kind of scene, see launch_persistent_t
    int tune_blocks_per_cu = 0, tune_batch = 0, tune_sdone_lanes = 2, tune_refill_lanes = 0, tune_max_leaf = 4, tune_tri_cost_x10 = 12, tune_lds_state = RT_LDS_PARK;

    void free_scene()
    {
        for (rt::f4 *p : {d_sph_leaf, d_nodes4q, d_nodes, d_nodes4, d_tri_isect, d_tri_shade, d_tri_shade_leaf, d_tri_raw, d_spheres, d_lights}) (void)hipFree(p);
        (void)hipFree(d_tri_orig);
        (void)hipFree(d_sph_orig);
        d_tri_orig = d_sph_orig = nullptr;
        d_sph_leaf = nullptr;
        d_nodes4q = d_nodes = d_nodes4 = d_tri_isect = d_tri_shade = d_tri_shade_leaf = d_tri_raw = d_spheres = d_lights = nullptr;
        has_scene = false;
    }
};

namespace {

constexpr uint32_t kWorkCounters = 64 * kLaunchCounters;     // ring of launches; far more than can be in flight
constexpr size_t kBatchBytesCap = (size_t)2 << 30;
constexpr uint64_t kUnitsPerLaunch = 64ull * 1920 * 1080;

// The debug view is drawn while the extension is switched on and the uniforms ask for it.
bool debug_view_active(const rt_context *c) { return c->debug_view && c->uniforms[17] != 0.0f; }

int launch_tiles(rt_context *c, const rt::FrameParams &fp, bool sum)
{
    dim3 block(kBlock);
    dim3 grid((c->width + 31) / 32, (c->local_rows + 7) / 8);
    if (grid.x == 0 || grid.y == 0) return RT_OK;
    uint32_t depth = c->host_scene.max_depth;
    rt::SceneView view = c->view;
    view.debug_view = debug_view_active(c) ? 1u : 0u;
#define RT_LAUNCH_A(BRUTE, STACK, COUNT)                                                                        \
    hipLaunchKernelGGL((render_tiles_kernel<BRUTE, STACK, COUNT>), grid, block, 0, c->stream, view, fp, c->sm, \
                       c->local_rows, c->fb, sum)
    if (c->mode == RT_MODE_BRUTE_FORCE || view.debug_view) { // the markers are in no tree: every primitive per ray
        RT_LAUNCH_A(true, 1, false);
    } else if (c->counting) {
        if (depth < 16) RT_LAUNCH_A(false, 16, true);
        else if (depth < 32) RT_LAUNCH_A(false, 32, true);
        else RT_LAUNCH_A(false, 64, true);
    } else {
        if (depth < 16) RT_LAUNCH_A(false, 16, false);
        else if (depth < 32) RT_LAUNCH_A(false, 32, false);
        else RT_LAUNCH_A(false, 64, false);
    }
#undef RT_LAUNCH_A
    RT_HIP_CHECK(hipGetLastError());
    return RT_OK;
}

int ensure_spill(rt_context *c, int which, hipStream_t stream, uint64_t blocks);

uint32_t tiles_per_frame_of(const rt_context *c)
{
    return (uint32_t)((c->width + 7) / 8) * (uint32_t)((c->local_rows + 7) / 8);
}

// Buffers of the tile hand-out order for stream `which` (0 main, 1.. side streams); costs start at zero.
int ensure_tile_order(rt_context *c, int which)
{
    size_t tiles = tiles_per_frame_of(c);
    if (tiles == 0 || c->d_tile_cls[which]) return RT_OK; // the last one made
    for (int k = 0; k < 2; k++)
        if (!c->d_tile_cost[which][k]) {
            RT_HIP_CHECK(hipMalloc((void **)&c->d_tile_cost[which][k], tiles * sizeof(unsigned int)));
            RT_HIP_CHECK(hipMemset(c->d_tile_cost[which][k], 0, tiles * sizeof(unsigned int)));
        }
    if (!c->d_tile_order[which]) RT_HIP_CHECK(hipMalloc((void **)&c->d_tile_order[which], tiles * sizeof(uint32_t)));
    // the zeroing above ran on the null stream, which the context's non-blocking streams do not wait for
    RT_HIP_CHECK(hipStreamSynchronize(nullptr));
    RT_HIP_CHECK(hipMalloc((void **)&c->d_tile_cls[which], tiles * sizeof(uint32_t)));
    return RT_OK;
}

template <class CFG, bool COUNT, bool FUSED>
int launch_persistent_t(rt_context *c, const rt::FrameParams &fp, const BatchParams &bp, unsigned int *counter, hipStream_t stream)
{
    auto kernel = render_persistent_kernel<CFG, COUNT, FUSED>;
    if (CFG::PLANE_LINKS == 16 && c->host_scene.links_packed != CFG::PLANE_LINKS)
        return fail(RT_ERR_UNSUPPORTED, "internal: kernel variant and node link packing do not match");
    if (!CFG::SPHERES && c->host_scene.n_tree_spheres != 0)
        return fail(RT_ERR_UNSUPPORTED, "internal: kernel variant without the sphere test for a scene whose tree holds spheres");
    int per_cu = c->tune_blocks_per_cu;
    if (per_cu <= 0) {
        RT_HIP_CHECK(hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, kernel, kBlock, 0));
        if (per_cu < 1) per_cu = 1;
    }
    uint64_t tiles = (uint64_t)((c->width + 7) / 8) * (uint64_t)((c->local_rows + 7) / 8) * bp.n_frames;
    uint64_t blocks = std::min<uint64_t>((uint64_t)c->n_cus * per_cu, (tiles + 3) / 4);
    if (blocks == 0) return RT_OK;
    if (CFG::SPILL) {
        // one column of kSpillEntries per lane of the grid; launches that share it are serialised by their
        // streams only within a stream, so every stream (main, side streams) has its own buffer.  The main
        // stream's is sized when the scene is uploaded (rt_render_frames_noalloc depends on that).
        int rc = ensure_spill(c, c->stream_index(stream), stream, blocks);
        if (rc) return rc;
    }
    int32_t *spill = CFG::SPILL ? c->d_spill[c->stream_index(stream)] : nullptr;
    if (COUNT) {
        size_t need = (size_t)blocks * (kBlock / 64) * 4;
        if (c->wave_trace_len < need) {
            RT_HIP_CHECK(hipStreamSynchronize(stream));
            (void)hipFree(c->fb.wave_trace);
            c->fb.wave_trace = nullptr;
            c->wave_trace_len = 0;
            RT_HIP_CHECK(hipMalloc((void **)&c->fb.wave_trace, need * sizeof(unsigned long long)));
            c->wave_trace_len = need;
        }
        c->wave_trace_used = need;
    }
    // Starting a pixel (seed, jitter, camera ray) is long code: with few samples per pixel it runs
    // often, so more idle lanes are collected before a refill (measured optimum per spp).
    int refill = c->tune_refill_lanes;
    if (refill <= 0) refill = fp.spp <= 2u ? 16 : fp.spp <= 8u ? 8 : 4;
    // Scenes of many triangles shade a little earlier and collect more lanes before a triangle step (each measured on
    // the 131k-triangle height field and on 991k triangles unseen / seen: 15.9 / 10.3 / 12.4 -> 15.2 / 9.7 / 12.3 ms/frame;
    // the small scenes lose 3 % with these values and keep theirs).
    const int shade_num = c->tune_shade_num >= 0 ? c->tune_shade_num : CFG::LINK16 ? 2 : 3;
    const int leaf_lanes = c->tune_leaf_lanes > 0 ? c->tune_leaf_lanes : CFG::LINK16 ? 12 : 16;
    SchedParams sched{shade_num, leaf_lanes, c->tune_sdone_lanes, refill, c->tune_shade_wait, c->tune_migrate};
    BatchParams bpl = bp;
    const int which = c->stream_index(stream);
    // (not for frames of a few samples per pixel: their units are short, so is the end of their launches, and the
    // bookkeeping costs more than it saves: +28 % on single 1-spp frames)
    if (c->tune_tile_order && c->d_tile_cls[which] && fp.spp >= 4u) {
        // the tiles that cost most in this stream's previous launch go first; this launch's costs go to the other buffer
        const uint32_t tiles_per_frame = tiles_per_frame_of(c);
        unsigned int *prev = c->d_tile_cost[which][c->tile_cost_phase[which]], *cur = c->d_tile_cost[which][c->tile_cost_phase[which] ^ 1];
        c->tile_cost_phase[which] ^= 1;
        RT_HIP_CHECK(hipMemsetAsync(cur, 0, (size_t)tiles_per_frame * sizeof(unsigned int), stream));
        hipLaunchKernelGGL(order_tiles_kernel, dim3(kQueues), dim3(256), 0, stream, prev, c->d_tile_cls[which], c->d_tile_order[which],
                           tiles_per_frame);
        RT_HIP_CHECK(hipGetLastError());
        bpl.tile_order = c->d_tile_order[which];
        bpl.tile_cost = cur;
    }
    hipLaunchKernelGGL(kernel, dim3((unsigned)blocks), dim3(kBlock), 0, stream, c->view, fp, c->sm, c->local_rows, c->fb,
                       bpl, counter, sched, spill);
    RT_HIP_CHECK(hipGetLastError());
    return RT_OK;
}

// Entries the persistent kernel's traversal stack must be able to hold for the current scene, and
// the most any instantiation provides.  A degenerate scene whose 4-wide tree needs more is rendered
// by the one-thread-per-pixel kernel instead, which walks the binary tree (need <= its depth < 64).
constexpr uint32_t kMaxStackEntries = 64;
uint32_t persistent_stack_need(const rt_context *c)
{
    return rt::kWideBvh ? c->host_scene.max_stack4 : c->host_scene.max_depth;
}

using CfgSmall = LaneCfg<RT_PARK2_STACK, RT_PARK2_SLOTS, RT_PARK2_MINW, true, false>;
using CfgSmallDeep = LaneCfg<32, 14, RT_MINW, true, false>;
#ifndef RT_LARGE_NODES
// Node form of the large-scene kernels: 0 = the 128-byte nodes, 64 = the 64-byte ones (8-bit child boxes relative to
// the node's own).  Measured with 64: four loads per visit instead of seven and half the node bytes, but 35 more
// VALU instructions per visit, and a plane can sit a whole step (1/255 of the parent's extent) further out: the
// floor and the walls, thin leaves directly under the root, then contain the origin of every ray that leaves them and
// are tested once more per ray (+0.5 to +1.7 triangle tests per ray).  991k triangles 11.9 -> 12.8 ms/frame, the
// unseen mesh 10.0 -> 12.0, the 131k-triangle height field 15.9 -> 17.9: not used.
#define RT_LARGE_NODES 0
#endif
using CfgLarge = LaneCfg<RT_LARGE_STACK, RT_LARGE_SLOTS, RT_LARGE_MINW, false, false, RT_LARGE_NODES>;
using CfgLargeDeep = LaneCfg<RT_LARGE_STACK, RT_LARGE_SLOTS, RT_LARGE_MINW, false, true, RT_LARGE_NODES>;
using CfgRegisters = LaneCfg<64, 0, 2, false, false, 0, false, true>; // 64 KB of LDS per block: two blocks per CU
// the sphere-light extension on small scenes: 22 x 2 + (14 + 7) x 4 = 128 bytes per lane, five blocks per CU
using CfgSmallSL = LaneCfg<RT_PARK2_STACK, 14, 5, true, false, RT_PACKED_LINKS ? 16 : 0, true, true>;
// scenes of many spheres, whose spheres are leaves of the tree: CfgSmallDeep and CfgLargeDeep with the sphere test in
using CfgSmallSpheres = LaneCfg<32, 14, RT_MINW, true, false, RT_PACKED_LINKS ? 16 : 0, false, true>;
using CfgLargeSpheres = LaneCfg<RT_LARGE_STACK, RT_LARGE_SLOTS, RT_LARGE_MINW, false, true, RT_LARGE_NODES, false, true>;

// Scenes whose links fit 16 bits: node indices below 32768 and leaf codes (first << 3 | count - 1) below 32768.
bool scene_is_small(const rt_context *c)
{
    const rt::BuiltScene &hs = c->host_scene;
    return hs.n_tris + hs.n_tree_spheres <= 4096 && hs.nodes.size() / 4 < 32768 && hs.nodes4.size() / 8 < 32768 && persistent_stack_need(c) <= 32 &&
           (hs.links_packed == 16 || !RT_PACKED_LINKS);
}

// The sphere-light extension runs on the persistent kernel for scenes of 16-bit links whose tree fits the stack of
// its instantiation; other scenes with the extension on go to the one-thread-per-pixel kernel.
bool sphere_lights_on_persistent(const rt_context *c)
{
    return rt::kWideBvh && scene_is_small(c) && persistent_stack_need(c) <= (uint32_t)CfgSmallSL::STACK;
}

// Whether the persistent kernel renders the current scene with the instantiation that keeps the deep end of
// its traversal stack in global memory.
bool scene_spills(const rt_context *c)
{
    return rt::kWideBvh && c->tune_lds_state != 0 && !scene_is_small(c) && persistent_stack_need(c) <= kMaxStackEntries &&
           (persistent_stack_need(c) > (uint32_t)CfgLarge::STACK || c->host_scene.n_tree_spheres != 0);
}

// Makes the stack overflow area of stream `which` (0 main, 1.. side streams) hold a column per lane of `blocks` blocks.
int ensure_spill(rt_context *c, int which, hipStream_t stream, uint64_t blocks)
{
    size_t need = (size_t)blocks * kBlock * kSpillEntries * sizeof(int32_t);
    if (c->spill_bytes[which] >= need) return RT_OK;
    if (c->no_alloc)
        return fail(RT_ERR_UNSUPPORTED, "rt_render_frames_noalloc: the stack overflow area in global memory is smaller than this launch "
                                        "needs (RT_TUNE_BLOCKS_PER_CU was raised after rt_set_scene sized it), and it cannot allocate");
    if (stream) RT_HIP_CHECK(hipStreamSynchronize(stream));
    (void)hipFree(c->d_spill[which]);
    c->d_spill[which] = nullptr;
    c->spill_bytes[which] = 0;
    RT_HIP_CHECK(hipMalloc((void **)&c->d_spill[which], need));
    c->spill_bytes[which] = need;
    return RT_OK;
}

// The most blocks any launch of the spilling instantiation puts on the device.
int spill_blocks_max(rt_context *c, uint64_t &blocks)
{
    int most = c->tune_blocks_per_cu;
    if (most <= 0) {
        int n[8] = {};
        RT_HIP_CHECK(hipOccupancyMaxActiveBlocksPerMultiprocessor(&n[0], render_persistent_kernel<CfgLargeDeep, false, false>, kBlock, 0));
        RT_HIP_CHECK(hipOccupancyMaxActiveBlocksPerMultiprocessor(&n[1], render_persistent_kernel<CfgLargeDeep, false, true>, kBlock, 0));
        RT_HIP_CHECK(hipOccupancyMaxActiveBlocksPerMultiprocessor(&n[2], render_persistent_kernel<CfgLargeDeep, true, false>, kBlock, 0));
        RT_HIP_CHECK(hipOccupancyMaxActiveBlocksPerMultiprocessor(&n[3], render_persistent_kernel<CfgLargeDeep, true, true>, kBlock, 0));
        RT_HIP_CHECK(hipOccupancyMaxActiveBlocksPerMultiprocessor(&n[4], render_persistent_kernel<CfgLargeSpheres, false, false>, kBlock, 0));
        RT_HIP_CHECK(hipOccupancyMaxActiveBlocksPerMultiprocessor(&n[5], render_persistent_kernel<CfgLargeSpheres, false, true>, kBlock, 0));
        RT_HIP_CHECK(hipOccupancyMaxActiveBlocksPerMultiprocessor(&n[6], render_persistent_kernel<CfgLargeSpheres, true, false>, kBlock, 0));
        RT_HIP_CHECK(hipOccupancyMaxActiveBlocksPerMultiprocessor(&n[7], render_persistent_kernel<CfgLargeSpheres, true, true>, kBlock, 0));
        most = 1;
        for (int v : n) most = std::max(most, v);
    }
    blocks = (uint64_t)c->n_cus * most;
    return RT_OK;
}

template <bool COUNT, bool FUSED>
int launch_persistent_d(rt_context *c, const rt::FrameParams &fp, const BatchParams &bp, unsigned int *counter, hipStream_t stream)
{
    // entries the traversal stack must hold
    uint32_t depth = persistent_stack_need(c);
    if (depth > kMaxStackEntries) return fail(RT_ERR_UNSUPPORTED, "acceleration structure too deep for the traversal stack");
    if ((c->tune_lds_state == 0 && !c->sphere_lights) || !rt::kWideBvh) return launch_persistent_t<CfgRegisters, COUNT, FUSED>(c, fp, bp, counter, stream);
    bool small = scene_is_small(c);
    if (c->sphere_lights) {
        if (!sphere_lights_on_persistent(c)) return fail(RT_ERR_UNSUPPORTED, "internal: sphere lights on a scene the persistent kernel does not take");
        return launch_persistent_t<CfgSmallSL, COUNT, FUSED>(c, fp, bp, counter, stream);
    }
    const bool tree_spheres = c->host_scene.n_tree_spheres != 0; // leaves of spheres: the variants with the sphere test in the triangle step
    if (small) {
        if (tree_spheres) return launch_persistent_t<CfgSmallSpheres, COUNT, FUSED>(c, fp, bp, counter, stream);
        if (depth <= (uint32_t)CfgSmall::STACK && c->tune_lds_state >= 2) return launch_persistent_t<CfgSmall, COUNT, FUSED>(c, fp, bp, counter, stream);
        return launch_persistent_t<CfgSmallDeep, COUNT, FUSED>(c, fp, bp, counter, stream);
    }
    if (tree_spheres) return launch_persistent_t<CfgLargeSpheres, COUNT, FUSED>(c, fp, bp, counter, stream);
    if (depth <= (uint32_t)CfgLarge::STACK) return launch_persistent_t<CfgLarge, COUNT, FUSED>(c, fp, bp, counter, stream);
    return launch_persistent_t<CfgLargeDeep, COUNT, FUSED>(c, fp, bp, counter, stream);
}

// Side streams, events and staging slots of the overlapped single-frame path, made on first use.
int ensure_slots(rt_context *c)
{
    size_t local_px = (size_t)c->local_rows * c->width, px = (size_t)c->width * c->height;
    for (int k = 0; k < c->tune_overlap && k < rt_context::kMaxSlots; k++) {
        if (c->d_slot_ids[k]) continue; // the last resource made: a slot whose creation failed half way is completed here
        if (!c->aux[k]) RT_HIP_CHECK(hipStreamCreateWithFlags(&c->aux[k], hipStreamNonBlocking));
        if (!c->trace_done[k]) RT_HIP_CHECK(hipEventCreateWithFlags(&c->trace_done[k], hipEventDisableTiming));
        if (!c->slot_free[k]) RT_HIP_CHECK(hipEventCreateWithFlags(&c->slot_free[k], hipEventDisableTiming));
        if (!c->d_slot[k]) RT_HIP_CHECK(hipMalloc((void **)&c->d_slot[k], std::max<size_t>(local_px, 1) * sizeof(uint2)));
        int rc = ensure_tile_order(c, 1 + k);
        if (rc) return rc;
        RT_HIP_CHECK(hipMalloc((void **)&c->d_slot_ids[k], px * sizeof(int32_t)));
    }
    return RT_OK;
}

// Frames per launch of rt_render_frames.  Every launch ends with a tail in which lanes finish their
// last pixel (16 serial samples, a few ms) while others idle, so launches should be long: by default
// up to kUnitsPerLaunch (frame, pixel) units, i.e. 64 frames of a 1920x1080 image, more frames when
// the context owns fewer pixels.  Bounded by the batch buffer size.
uint32_t frames_per_launch(const rt_context *c)
{
    size_t local_px = std::max<size_t>((size_t)c->local_rows * c->width, 1);
    uint64_t tiles_per_frame = std::max<uint64_t>((uint64_t)((c->width + 7) / 8) * (uint64_t)((c->local_rows + 7) / 8), 1);
    uint64_t want = c->tune_batch > 0 ? (uint64_t)c->tune_batch
                                      : std::max<uint64_t>(1, (kUnitsPerLaunch + local_px / 2) / local_px);
    return (uint32_t)std::max<uint64_t>(1, std::min<uint64_t>(want, std::min<uint64_t>(kBatchBytesCap / (local_px * sizeof(uint2)),
                                                                                  0x7fffffffull / (tiles_per_frame * 64))));
}

// Frames first, first + stride, ... (n of them) through the persistent kernel, in batches.
int launch_persistent(rt_context *c, uint32_t first, uint32_t n, uint32_t stride, bool sum)
{
    size_t local_px = (size_t)c->local_rows * c->width;
    if (local_px == 0 || n == 0) return RT_OK;
    uint32_t max_batch = frames_per_launch(c);
    if (c->no_alloc) {
        // the workspace holds the frame colours of one launch
        if (local_px * sizeof(uint2) > kScratchColorBytes)
            return fail(RT_ERR_UNSUPPORTED, "rt_render_frames_noalloc: the context owns " + std::to_string(local_px) +
                                            " pixels, the fixed workspace holds " + std::to_string(kScratchColorBytes / sizeof(uint2)));
        max_batch = (uint32_t)std::min<uint64_t>(max_batch, kScratchColorBytes / (local_px * sizeof(uint2)));
    }
    uint32_t done = 0;
    while (done < n) {
        uint32_t b = std::min(max_batch, n - done);
        unsigned int *counter = c->d_work + c->work_used;
        c->work_used = (c->work_used + kLaunchCounters) % kWorkCounters;
        if (c->no_alloc) counter = (unsigned int *)c->scratch; // launches on one stream run in order: one counter is enough
        c->uniforms[12] = (float)(first + done * stride);
        rt::FrameParams fp;
        std::string err;
        if (!rt::make_frame_params(c->uniforms, c->max_bounces, fp, err)) return fail(RT_ERR_INVALID, err);
        BatchParams bp{first + done * stride, b, nullptr, stride, sum, c->fb.ids, false};
        dim3 acc_grid((unsigned)((local_px + 255) / 256));
        int rc;
        if (c->no_alloc) {
            // same kernels as the batch path below, colours in the fixed workspace
            bp.colors = (uint2 *)((unsigned char *)c->scratch + kScratchHeader);
            RT_HIP_CHECK(hipMemsetAsync(counter, 0, kLaunchCounters * sizeof(unsigned int), c->stream));
            rc = c->counting ? launch_persistent_d<true, false>(c, fp, bp, counter, c->stream)
                             : launch_persistent_d<false, false>(c, fp, bp, counter, c->stream);
            if (rc) return rc;
            hipLaunchKernelGGL(accumulate_batch_kernel, acc_grid, dim3(256), 0, c->stream, c->fb, bp, c->width, c->height,
                               c->local_rows, c->sm);
            RT_HIP_CHECK(hipGetLastError());
        } else if (b == 1 && !c->tune_overlap) {
            // one frame, accumulated by the trace kernel itself
            RT_HIP_CHECK(hipMemsetAsync(counter, 0, kLaunchCounters * sizeof(unsigned int), c->stream));
            rc = c->counting ? launch_persistent_d<true, true>(c, fp, bp, counter, c->stream)
                             : launch_persistent_d<false, true>(c, fp, bp, counter, c->stream);
            if (rc) return rc;
        } else if (b == 1) {
            // one frame, traced on a side stream into a staging slot, accumulated on the main stream
            if ((rc = ensure_slots(c))) return rc;
            int k = c->next_slot % c->tune_overlap;
            c->next_slot = (k + 1) % c->tune_overlap;
            // The slot is free once the accumulate step of the frame before last has read it.  Waiting for
            // that on the host is the back-pressure of a render loop that never syncs: at most two
            // frames are in flight.
            if (c->slot_used[k]) RT_HIP_CHECK(hipEventSynchronize(c->slot_free[k]));
            bp.colors = c->d_slot[k];
            bp.ids = c->d_slot_ids[k];
            bp.copy_ids = true;
            // rt_last_render_ms spans from the first trace to the last accumulate: restart its first event here
            if (done == 0) RT_HIP_CHECK(hipEventRecord(c->ev_start, c->aux[k]));
            RT_HIP_CHECK(hipMemsetAsync(counter, 0, kLaunchCounters * sizeof(unsigned int), c->aux[k]));
            rc = c->counting ? launch_persistent_d<true, false>(c, fp, bp, counter, c->aux[k])
                             : launch_persistent_d<false, false>(c, fp, bp, counter, c->aux[k]);
            if (rc) return rc;
            RT_HIP_CHECK(hipEventRecord(c->trace_done[k], c->aux[k]));
            RT_HIP_CHECK(hipStreamWaitEvent(c->stream, c->trace_done[k], 0));
            hipLaunchKernelGGL(accumulate_batch_kernel, acc_grid, dim3(256), 0, c->stream, c->fb, bp, c->width, c->height,
                               c->local_rows, c->sm);
            RT_HIP_CHECK(hipGetLastError());
            RT_HIP_CHECK(hipEventRecord(c->slot_free[k], c->stream));
            c->slot_used[k] = true;
        } else {
            if (c->batch_capacity < b) {
                if (c->d_batch) { RT_HIP_CHECK(hipStreamSynchronize(c->stream)); (void)hipFree(c->d_batch); c->d_batch = nullptr; c->batch_capacity = 0; }
                RT_HIP_CHECK(hipMalloc((void **)&c->d_batch, (size_t)b * local_px * sizeof(uint2)));
                c->batch_capacity = b;
            }
            bp.colors = c->d_batch;
            RT_HIP_CHECK(hipMemsetAsync(counter, 0, kLaunchCounters * sizeof(unsigned int), c->stream));
            rc = c->counting ? launch_persistent_d<true, false>(c, fp, bp, counter, c->stream)
                             : launch_persistent_d<false, false>(c, fp, bp, counter, c->stream);
            if (rc) return rc;
            hipLaunchKernelGGL(accumulate_batch_kernel, acc_grid, dim3(256), 0, c->stream, c->fb, bp, c->width, c->height,
                               c->local_rows, c->sm);
            RT_HIP_CHECK(hipGetLastError());
        }
        done += b;
    }
    c->uniforms[12] = (float)(first + (n - 1) * stride);
    return RT_OK;
}

// The staging block of rt_read_owned_rows / rt_write_rows, kept between calls (both calls end with a
// stream synchronisation, so it is idle whenever one of them starts).
int ensure_row_staging(rt_context *c, size_t bytes)
{
    if (c->row_staging_bytes >= bytes) return RT_OK;
    RT_HIP_CHECK(hipStreamSynchronize(c->stream));
    (void)hipFree(c->d_row_staging);
    c->d_row_staging = nullptr;
    c->row_staging_bytes = 0;
    RT_HIP_CHECK(hipMalloc((void **)&c->d_row_staging, bytes));
    c->row_staging_bytes = bytes;
    return RT_OK;
}

int check_ready(rt_context *c)
{
    if (!c) return fail(RT_ERR_INVALID, "null context");
    if (!c->has_scene) return fail(RT_ERR_INVALID, "rt_set_scene has not been called");
    if (!c->has_uniforms) return fail(RT_ERR_INVALID, "rt_set_uniforms has not been called");
    return RT_OK;
}

} // namespace

static int render_frames_impl(rt_context *c, uint32_t first_frame_index, uint32_t n_frames, uint32_t stride, bool sum);

// Whether frames go through the persistent kernel: the default mode without the extensions that only the
// one-thread-per-pixel kernels implement, a tree its stack can hold, and no more samples per pixel-frame than
// its lanes count (4096; the reference's own limit is 16, src/webgpu-renderer.ts:233).
static bool persistent_path(const rt_context *c)
{
    return c->mode == RT_MODE_BVH && (!c->sphere_lights || sphere_lights_on_persistent(c)) && !debug_view_active(c) &&
           persistent_stack_need(c) <= (uint32_t)c->tune_max_stack && c->uniforms[16] <= (float)rt::kMaxLaneSamples;
}

// Submission coalescing.  rt_render_frame is the reference's fire-and-forget submit: one frame per
// call, called back to back by a render loop.  A launch of one frame ends with a tail in which lanes
// finish their last pixel while the others idle, so frames submitted while the GPU is still busy
// with earlier ones are held back (up to tune_coalesce of them, consecutive frame indices, same
// uniforms otherwise) and launched together as one batch, which is what rt_render_frames does.
// Frames are launched at once while fewer than two launches are in flight, so a GPU that has
// nothing to do never waits for a batch to fill.  Every other entry point launches what is pending
// before it does anything else, so the order of effects is the order of the calls.
static int flush_pending(rt_context *c)
{
    if (c->pending_n == 0) return RT_OK;
    uint32_t first = c->pending_first, n = c->pending_n;
    c->pending_n = 0;
    int rc = render_frames_impl(c, first, n, 1, false);
    if (rc) return rc;
    int k = c->next_flush;
    c->next_flush ^= 1;
    if (!c->flush_done[k]) RT_HIP_CHECK(hipEventCreateWithFlags(&c->flush_done[k], hipEventDisableTiming));
    RT_HIP_CHECK(hipEventRecord(c->flush_done[k], c->stream));
    c->flush_used[k] = true;
    return RT_OK;
}

// Launches made by flush_pending that have not finished yet (0..2).
static int launches_in_flight(rt_context *c)
{
    int n = 0;
    for (int k = 0; k < 2; k++)
        if (c->flush_used[k]) {
            if (hipEventQuery(c->flush_done[k]) == hipErrorNotReady) n++;
            else c->flush_used[k] = false;
        }
    return n;
}

#define RT_FLUSH_PENDING(c)                     \
    do {                                        \
        int frc_ = flush_pending(c);            \
        if (frc_) return frc_;                  \
    } while (0)

extern "C" {

const char *rt_last_error(void) { return g_last_error.c_str(); }

int rt_device_count(void)
{
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) return 0;
    return n;
}

int rt_create(int device, int width, int height, int stripe_rows, int stripe_index, int stripe_count, rt_context **out)
{
    if (!out) return fail(RT_ERR_INVALID, "out is null");
    *out = nullptr;
    if (width < 1 || height < 1 || width > 65536 || height > 65536) return fail(RT_ERR_INVALID, "width/height out of range");
    if (stripe_count < 1 || stripe_index < 0 || stripe_index >= stripe_count || stripe_rows < 1)
        return fail(RT_ERR_INVALID, "bad stripe assignment");
    // work units of a launch are numbered in 32 bits: one frame of the image, padded to 8x8 tiles, must fit
    if ((uint64_t)((width + 7) / 8) * (uint64_t)((height + 7) / 8) * 64u >= (1ull << 31))
        return fail(RT_ERR_UNSUPPORTED, "image larger than 2^31 pixels");
    int n_dev = 0;
    RT_HIP_CHECK(hipGetDeviceCount(&n_dev));
    if (device < 0 || device >= n_dev) return fail(RT_ERR_DEVICE, "no such HIP device");
    RT_HIP_CHECK(hipSetDevice(device));

    rt_context *c = new (std::nothrow) rt_context();
    if (!c) return fail(RT_ERR_DEVICE, "out of host memory");
    c->device = device;
    c->width = width;
    c->height = height;
    c->sm = StripeMap{stripe_rows, stripe_index, stripe_count};
    c->local_rows = owned_rows_of(height, c->sm);

    size_t px = (size_t)width * height;
    auto cleanup = [&](int code) { rt_destroy(c); return code; };
    hipError_t e;
    if ((e = hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking)) != hipSuccess ||
        (e = hipEventCreate(&c->ev_start)) != hipSuccess || (e = hipEventCreate(&c->ev_stop)) != hipSuccess ||
        (e = hipMalloc((void **)&c->fb.accum, px * 3 * sizeof(float))) != hipSuccess ||
        (e = hipMalloc((void **)&c->fb.frame_f16, px * 4 * sizeof(uint16_t))) != hipSuccess ||
        (e = hipMalloc((void **)&c->fb.ids, px * sizeof(int32_t))) != hipSuccess ||
        (e = hipMalloc((void **)&c->fb.counters, 32 * sizeof(unsigned long long))) != hipSuccess ||
        (e = hipMalloc((void **)&c->d_work, kWorkCounters * sizeof(unsigned int))) != hipSuccess ||
        (e = hipDeviceGetAttribute(&c->n_cus, hipDeviceAttributeMultiprocessorCount, device)) != hipSuccess ||
        (e = hipMemsetAsync(c->fb.accum, 0, px * 3 * sizeof(float), c->stream)) != hipSuccess ||
        (e = hipMemsetAsync(c->fb.frame_f16, 0, px * 4 * sizeof(uint16_t), c->stream)) != hipSuccess ||
        (e = hipMemsetAsync(c->fb.counters, 0, 32 * sizeof(unsigned long long), c->stream)) != hipSuccess) {
        return cleanup(fail(RT_ERR_DEVICE, std::string("rt_create: ") + hipGetErrorString(e)));
    }
    hipLaunchKernelGGL(fill_i32_kernel, dim3((unsigned)((px + 255) / 256)), dim3(256), 0, c->stream, c->fb.ids, -2, px);
    if ((e = hipStreamSynchronize(c->stream)) != hipSuccess)
        return cleanup(fail(RT_ERR_DEVICE, std::string("rt_create: ") + hipGetErrorString(e)));
    {
        int rc = ensure_tile_order(c, 0); // here, not at the first launch: rt_render_frames_noalloc must find it made
        if (rc) return cleanup(rc);
    }
    *out = c;
    return RT_OK;
}

void rt_destroy(rt_context *c)
{
    if (!c) return;
    (void)hipSetDevice(c->device);
    if (c->stream) (void)hipStreamSynchronize(c->stream);
    c->free_scene();
    for (int k = 0; k < rt_context::kMaxSlots; k++) {
        if (c->aux[k]) { (void)hipStreamSynchronize(c->aux[k]); (void)hipStreamDestroy(c->aux[k]); }
        if (c->trace_done[k]) (void)hipEventDestroy(c->trace_done[k]);
        if (c->slot_free[k]) (void)hipEventDestroy(c->slot_free[k]);
        (void)hipFree(c->d_slot[k]); (void)hipFree(c->d_slot_ids[k]);
    }
    for (hipEvent_t e : c->flush_done)
        if (e) (void)hipEventDestroy(e);
    if (c->copy_stream) { (void)hipStreamSynchronize(c->copy_stream); (void)hipStreamDestroy(c->copy_stream); }
    for (int k = 0; k < 2; k++) {
        (void)hipFree(c->d_present[k]);
        if (c->h_present[k]) (void)hipHostFree(c->h_present[k]);
        if (c->present_ready[k]) (void)hipEventDestroy(c->present_ready[k]);
        if (c->present_done[k]) (void)hipEventDestroy(c->present_done[k]);
    }
    for (int32_t *p : c->d_spill) (void)hipFree(p);
    for (int k = 0; k <= rt_context::kMaxSlots; k++) {
        (void)hipFree(c->d_tile_cost[k][0]); (void)hipFree(c->d_tile_cost[k][1]);
        (void)hipFree(c->d_tile_order[k]); (void)hipFree(c->d_tile_cls[k]);
    }
    (void)hipFree(c->d_work); (void)hipFree(c->d_batch); (void)hipFree(c->d_row_staging); (void)hipFree(c->fb.wave_trace);
    (void)hipFree(c->fb.accum); (void)hipFree(c->fb.frame_f16); (void)hipFree(c->fb.ids); (void)hipFree(c->fb.counters);
    if (c->ev_start) (void)hipEventDestroy(c->ev_start);
    if (c->ev_stop) (void)hipEventDestroy(c->ev_stop);
    if (c->stream) (void)hipStreamDestroy(c->stream);
    delete c;
}

// Builds the acceleration structure and the tables on the host and uploads them.  origin_mag: see
// rt::BuildOptions (the camera's coordinate magnitude when it is far from the scene, else 0).
static int upload_scene(rt_context *c, const void *spheres, uint32_t n_spheres, const void *triangles, uint32_t n_triangles,
                        float origin_mag)
{
    RT_HIP_CHECK(hipSetDevice(c->device));
    RT_HIP_CHECK(hipStreamSynchronize(c->stream));
    for (hipStream_t s : c->aux)
        if (s) RT_HIP_CHECK(hipStreamSynchronize(s));
    c->free_scene();
    rt::BuildOptions opt;
    opt.max_leaf = (uint32_t)c->tune_max_leaf;
    opt.triangle_cost = 0.1f * (float)c->tune_tri_cost_x10;
    opt.origin_mag = origin_mag;
    opt.quant_nodes = RT_LARGE_NODES == 64;
    opt.linear_spheres = (uint32_t)c->tune_linear_spheres;
    rt::build_scene(spheres, n_spheres, triangles, n_triangles, opt, c->host_scene);
    if (c->host_scene.max_depth >= 64) return fail(RT_ERR_UNSUPPORTED, "acceleration structure too deep");
    // the 4-wide node step addresses nodes by a 32-bit byte offset
    if (c->host_scene.nodes4.size() * sizeof(rt::f4) >= (1ull << 32)) return fail(RT_ERR_UNSUPPORTED, "too many nodes");
    const rt::BuiltScene &hs = c->host_scene;
    int rc;
    if ((rc = upload(hs.nodes, &c->d_nodes, c->stream)) || (rc = upload(hs.nodes4, &c->d_nodes4, c->stream)) || (rc = upload(hs.nodes4q, &c->d_nodes4q, c->stream)) ||
        (rc = upload(hs.sph_leaf, &c->d_sph_leaf, c->stream)) || (rc = upload(hs.sph_orig, &c->d_sph_orig, c->stream)) || (rc = upload(hs.tri_isect, &c->d_tri_isect, c->stream)) ||
        (rc = upload(hs.tri_shade, &c->d_tri_shade, c->stream)) || (rc = upload(hs.tri_shade_leaf, &c->d_tri_shade_leaf, c->stream)) ||
        (rc = upload(hs.tri_orig, &c->d_tri_orig, c->stream)) || (rc = upload(hs.tri_raw, &c->d_tri_raw, c->stream)) ||
        (rc = upload(hs.spheres, &c->d_spheres, c->stream)) || (rc = upload(hs.lights, &c->d_lights, c->stream))) {
        c->free_scene();
        return rc;
    }
    RT_HIP_CHECK(hipStreamSynchronize(c->stream));
    c->scene_bytes = (hs.nodes.size() + hs.nodes4.size() + hs.nodes4q.size() + hs.tri_isect.size() + hs.tri_shade.size() + hs.tri_shade_leaf.size() + hs.tri_orig.size() / 4 + hs.spheres.size() + hs.lights.size()) * sizeof(rt::f4);
    rt::SceneView &v = c->view;
    v.nodes = c->d_nodes; v.nodes4 = c->d_nodes4; v.nodes4q = c->d_nodes4q; v.root4 = hs.root4; v.tri_isect = c->d_tri_isect; v.tri_shade = c->d_tri_shade; v.tri_shade_leaf = c->d_tri_shade_leaf; v.tri_orig = c->d_tri_orig; v.tri_raw = c->d_tri_raw;
    v.spheres = c->d_spheres; v.lights = c->d_lights;
    v.sph_leaf = c->d_sph_leaf; v.sph_orig = c->d_sph_orig; v.n_linear_spheres = hs.n_spheres - hs.n_tree_spheres;
    v.n_spheres = hs.n_spheres; v.n_tris = hs.n_tris; v.n_lights = hs.n_lights; v.n_lights_pass1 = hs.n_lights_pass1;
    v.total_power = hs.total_power; v.root = hs.root;
    v.debug_view = 0u;
    c->has_scene = true;
    if (scene_spills(c)) {
        // the overflow area of the context's own stream is made here, not at the first launch that needs it,
        // so that rt_render_frames_noalloc never depends on an earlier allocating call
        uint64_t blocks = 0;
        if ((rc = spill_blocks_max(c, blocks)) || (rc = ensure_spill(c, 0, nullptr, blocks))) return rc;
    }
    return RT_OK;
}

// Largest coordinate magnitude of the camera position in the stored uniforms (0 if not finite).
static float camera_magnitude(const rt_context *c)
{
    float m = std::max(std::fabs(c->uniforms[0]), std::max(std::fabs(c->uniforms[1]), std::fabs(c->uniforms[2])));
    return m < INFINITY ? m : 0.0f;
}

int rt_set_scene(rt_context *c, const void *spheres, uint32_t n_spheres, const void *triangles, uint32_t n_triangles)
{
    if (!c) return fail(RT_ERR_INVALID, "null context");
    RT_FLUSH_PENDING(c);
    if ((n_spheres && !spheres) || (n_triangles && !triangles)) return fail(RT_ERR_INVALID, "null scene buffer");
    if (n_triangles >= (1u << 28)) return fail(RT_ERR_UNSUPPORTED, "too many triangles");
    float cam = c->has_uniforms ? camera_magnitude(c) : 0.0f;
    return upload_scene(c, spheres, n_spheres, triangles, n_triangles, cam > rt::kOriginMagCovered ? 2.0f * cam : 0.0f);
}

// The boxes of the acceleration structure are padded for ray origins up to host_scene.origin_mag.  A
// camera further out than that (and than the fixed margin covers) gets the structure rebuilt with
// wider boxes, for twice its distance so that a moving camera does not rebuild every frame.
static int ensure_padding_covers_camera(rt_context *c)
{
    float cam = camera_magnitude(c);
    if (!(cam > rt::kOriginMagCovered) || !(cam > c->host_scene.origin_mag)) return RT_OK;
    if (c->no_alloc)
        return fail(RT_ERR_UNSUPPORTED, "the camera moved beyond the range the acceleration structure was built for; "
                                        "rebuilding it allocates, which rt_render_frames_noalloc must not");
    std::vector<rt::f4> spheres = c->host_scene.spheres, tris = c->host_scene.tri_raw;
    return upload_scene(c, spheres.data(), c->host_scene.n_spheres, tris.data(), c->host_scene.n_tris,
                        std::min(2.0f * cam, 1e30f));
}

int rt_set_uniforms(rt_context *c, const void *uniforms80)
{
    if (!c || !uniforms80) return fail(RT_ERR_INVALID, "null argument");
    rt::FrameParams fp;
    std::string err;
    if (!rt::make_frame_params(uniforms80, c->max_bounces, fp, err)) return fail(RT_ERR_INVALID, err);
    if (fp.width != c->width || fp.height != c->height)
        return fail(RT_ERR_INVALID, "uniforms.resolution differs from the context's width/height");
    if (c->pending_n) {
        // frames held back were asked for with the stored uniforms: they can stay pending only if
        // nothing but frameIndex changes
        float u[20];
        std::memcpy(u, uniforms80, 80);
        u[12] = c->uniforms[12];
        if (std::memcmp(u, c->uniforms, 80) != 0) RT_FLUSH_PENDING(c);
    }
    std::memcpy(c->uniforms, uniforms80, 80);
    c->has_uniforms = true;
    return RT_OK;
}

int rt_set_max_bounces(rt_context *c, uint32_t max_bounces)
{
    if (!c) return fail(RT_ERR_INVALID, "null context");
    RT_FLUSH_PENDING(c);
    if (max_bounces > 255u) return fail(RT_ERR_INVALID, "max_bounces must be at most 255");
    c->max_bounces = max_bounces;
    return RT_OK;
}

int rt_set_sphere_lights(rt_context *c, int enable)
{
    if (!c) return fail(RT_ERR_INVALID, "null context");
    RT_FLUSH_PENDING(c);
    c->sphere_lights = enable != 0;
    return RT_OK;
}

int rt_set_debug_view(rt_context *c, int enable)
{
    if (!c) return fail(RT_ERR_INVALID, "null context");
    RT_FLUSH_PENDING(c);
    c->debug_view = enable != 0;
    return RT_OK;
}

int rt_set_mode(rt_context *c, int mode)
{
    if (!c) return fail(RT_ERR_INVALID, "null context");
    RT_FLUSH_PENDING(c);
    if (mode != RT_MODE_BVH && mode != RT_MODE_BRUTE_FORCE && mode != RT_MODE_BVH_TILES) return fail(RT_ERR_INVALID, "unknown mode");
    c->mode = mode;
    return RT_OK;
}

int rt_set_tuning(rt_context *c, int key, int value)
{
    if (!c) return fail(RT_ERR_INVALID, "null context");
    RT_FLUSH_PENDING(c);
    switch (key) {
    case RT_TUNE_SHADE_NUM: if (value < 0 || value > 8) return fail(RT_ERR_INVALID, "shade_num out of range"); c->tune_shade_num = value; break;
    case RT_TUNE_LEAF_LANES: if (value < 1 || value > 64) return fail(RT_ERR_INVALID, "leaf_lanes out of range"); c->tune_leaf_lanes = value; break;
    case RT_TUNE_BLOCKS_PER_CU: if (value < 0 || value > 64) return fail(RT_ERR_INVALID, "blocks_per_cu out of range"); c->tune_blocks_per_cu = value; break;
    case RT_TUNE_BATCH_FRAMES: if (value < 0 || value > 4096) return fail(RT_ERR_INVALID, "batch_frames out of range"); c->tune_batch = value; break;
    case RT_TUNE_MAX_LEAF: if (value < 1 || value > 8) return fail(RT_ERR_INVALID, "max_leaf out of range"); c->tune_max_leaf = value; break;
    case RT_TUNE_TRI_COST_X10: if (value < 1 || value > 1000) return fail(RT_ERR_INVALID, "tri_cost out of range"); c->tune_tri_cost_x10 = value; break;
    case RT_TUNE_REFILL_LANES: if (value < 0 || value > 64) return fail(RT_ERR_INVALID, "refill_lanes out of range"); c->tune_refill_lanes = value; break;
    case RT_TUNE_OVERLAP_FRAMES:
        if (value < 0 || value > rt_context::kMaxSlots) return fail(RT_ERR_INVALID, "overlap_frames out of range");
        // frames still in flight on side streams that the new setting would no longer wait for
        RT_HIP_CHECK(hipSetDevice(c->device));
        RT_HIP_CHECK(hipStreamSynchronize(c->stream));
        c->tune_overlap = value == 1 ? 2 : value; // 1 = the original two side streams
        break;
    case RT_TUNE_SHADE_WAIT: if (value < 0 || value > 1000) return fail(RT_ERR_INVALID, "shade_wait out of range"); c->tune_shade_wait = value; break;
    case RT_TUNE_LDS_STATE: if (value < 0 || value > 2) return fail(RT_ERR_INVALID, "lds_state out of range"); c->tune_lds_state = value; break;
    case RT_TUNE_COALESCE_FRAMES: if (value < 0 || value > 256) return fail(RT_ERR_INVALID, "coalesce_frames out of range"); c->tune_coalesce = value; break;
    case RT_TUNE_MAX_STACK: if (value < 1 || value > (int)kMaxStackEntries) return fail(RT_ERR_INVALID, "max_stack out of range"); c->tune_max_stack = value; break;
    case RT_TUNE_SHADOW_DONE_LANES: if (value < 1 || value > 64) return fail(RT_ERR_INVALID, "shadow_done_lanes out of range"); c->tune_sdone_lanes = value; break;
    case RT_TUNE_LINEAR_SPHERES: if (value < 0) return fail(RT_ERR_INVALID, "linear_spheres out of range"); c->tune_linear_spheres = value; break;
    case RT_TUNE_TILE_ORDER: if (value < 0 || value > 1) return fail(RT_ERR_INVALID, "tile_order must be 0 or 1"); c->tune_tile_order = value; break;
    case RT_TUNE_MIGRATE_LANES: if (value < 0 || value > 64) return fail(RT_ERR_INVALID, "migrate_lanes out of range"); c->tune_migrate = RT_MIGRATE ? value : 0; break;
    default: return fail(RT_ERR_INVALID, "unknown tuning key");
    }
    return RT_OK;
}

static int render_frames_impl(rt_context *c, uint32_t first_frame_index, uint32_t n_frames, uint32_t stride, bool sum)
{
    int rc = check_ready(c);
    if (rc) return rc;
    if (stride < 1) return fail(RT_ERR_INVALID, "frame stride must be at least 1");
    if (n_frames && (uint64_t)first_frame_index + (uint64_t)(n_frames - 1) * stride >= (1ull << 24))
        return fail(RT_ERR_INVALID, "frame index beyond 2^24 is not exact in the f32 uniform");
    RT_HIP_CHECK(hipSetDevice(c->device));
    if ((rc = ensure_padding_covers_camera(c))) return rc;
    RT_HIP_CHECK(hipEventRecord(c->ev_start, c->stream));
    if (persistent_path(c)) {
        if ((rc = launch_persistent(c, first_frame_index, n_frames, stride, sum))) return rc;
    } else {
        for (uint32_t k = 0; k < n_frames; k++) {
            c->uniforms[12] = (float)(first_frame_index + k * stride);
            rt::FrameParams fp;
            std::string err;
            if (!rt::make_frame_params(c->uniforms, c->max_bounces, fp, err)) return fail(RT_ERR_INVALID, err);
            fp.sphere_lights = c->sphere_lights ? 1u : 0u;
            if ((rc = launch_tiles(c, fp, sum))) return rc;
        }
    }
    RT_HIP_CHECK(hipEventRecord(c->ev_stop, c->stream));
    c->timed = true;
    c->timed_frames = n_frames;
    if (n_frames) { c->launch_serial++; c->launched_frame = first_frame_index + (n_frames - 1) * stride; }
    return RT_OK;
}

int rt_render_frames(rt_context *c, uint32_t first_frame_index, uint32_t n_frames)
{
    if (!c) return fail(RT_ERR_INVALID, "null context");
    RT_FLUSH_PENDING(c);
    return render_frames_impl(c, first_frame_index, n_frames, 1, false);
}

int rt_render_frames_noalloc(rt_context *c, uint32_t first_frame_index, uint32_t n_frames)
{
    if (!c) return fail(RT_ERR_INVALID, "null context");
    if (!c->scratch) {
        RT_HIP_CHECK(hipSetDevice(c->device));
        int rc = scratch_of(c->device, &c->scratch);
        if (rc) return rc;
    }
    c->no_alloc = true;
    int rc = flush_pending(c); // frames held back by rt_render_frame go first, through the same workspace
    if (!rc) rc = render_frames_impl(c, first_frame_index, n_frames, 1, false);
    c->no_alloc = false;
    return rc;
}

int rt_render_frame_sums(rt_context *c, uint32_t first_frame_index, uint32_t n_frames, uint32_t stride)
{
    if (!c) return fail(RT_ERR_INVALID, "null context");
    RT_FLUSH_PENDING(c);
    return render_frames_impl(c, first_frame_index, n_frames, stride, true);
}

int rt_clear_accum(rt_context *c)
{
    if (!c) return fail(RT_ERR_INVALID, "null context");
    RT_FLUSH_PENDING(c);
    RT_HIP_CHECK(hipSetDevice(c->device));
    RT_HIP_CHECK(hipMemsetAsync(c->fb.accum, 0, (size_t)c->width * c->height * 3 * sizeof(float), c->stream));
    c->launch_serial++; // the planes changed: the next rt_present makes a new image
    return RT_OK;
}

int rt_export_accum(rt_context *c, void *dst_device)
{
    if (!c || !dst_device) return fail(RT_ERR_INVALID, "null argument");
    RT_FLUSH_PENDING(c);
    RT_HIP_CHECK(hipSetDevice(c->device));
    RT_HIP_CHECK(hipMemcpyAsync(dst_device, c->fb.accum, (size_t)c->width * c->height * 3 * sizeof(float),
                                hipMemcpyDeviceToDevice, c->stream));
    return RT_OK;
}

int rt_import_accum(rt_context *c, const void *src_device)
{
    if (!c || !src_device) return fail(RT_ERR_INVALID, "null argument");
    RT_FLUSH_PENDING(c);
    RT_HIP_CHECK(hipSetDevice(c->device));
    RT_HIP_CHECK(hipMemcpyAsync(c->fb.accum, src_device, (size_t)c->width * c->height * 3 * sizeof(float),
                                hipMemcpyDeviceToDevice, c->stream));
    c->launch_serial++; // the planes changed: the next rt_present makes a new image
    return RT_OK;
}

int rt_finish_frame_sums(rt_context *c, uint32_t n_frames_total)
{
    if (!c) return fail(RT_ERR_INVALID, "null context");
    RT_FLUSH_PENDING(c);
    RT_HIP_CHECK(hipSetDevice(c->device));
    size_t n = (size_t)c->width * c->height * 3;
    hipLaunchKernelGGL(finish_sums_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, c->stream, c->fb.accum, n,
                       (float)n_frames_total + 1.0f);
    RT_HIP_CHECK(hipGetLastError());
    c->launch_serial++; // the planes changed: the next rt_present makes a new image
    return RT_OK;
}

int rt_render_frame(rt_context *c)
{
    int rc = check_ready(c);
    if (rc) return rc;
    uint32_t frame = (uint32_t)c->uniforms[12];
    if (frame >= (1u << 24)) return fail(RT_ERR_INVALID, "frame index beyond 2^24 is not exact in the f32 uniform");
    bool coalesce = c->tune_coalesce > 1 && !c->counting && persistent_path(c);
    if (!coalesce) {
        RT_FLUSH_PENDING(c);
        return render_frames_impl(c, frame, 1, 1, false);
    }
    if (c->pending_n && frame != c->pending_first + c->pending_n) RT_FLUSH_PENDING(c);
    if (c->pending_n == 0) c->pending_first = frame;
    c->pending_n++;
    // Launch now if the GPU would otherwise run dry (fewer than two launches in flight); hold the frame
    // back otherwise, up to tune_coalesce frames, then wait for the older launch (the back-pressure of a
    // render loop that never syncs) and launch them as one batch.
    RT_HIP_CHECK(hipSetDevice(c->device));
    int busy = launches_in_flight(c);
    if (busy >= 2 && c->pending_n >= (uint32_t)c->tune_coalesce) {
        int older = c->next_flush; // the slot flush_pending records into next holds the older launch
        if (c->flush_used[older]) RT_HIP_CHECK(hipEventSynchronize(c->flush_done[older]));
        busy = launches_in_flight(c);
    }
    if (busy < 2) RT_FLUSH_PENDING(c);
    return RT_OK;
}

int rt_sync(rt_context *c)
{
    if (!c) return fail(RT_ERR_INVALID, "null context");
    RT_FLUSH_PENDING(c);
    RT_HIP_CHECK(hipSetDevice(c->device));
    RT_HIP_CHECK(hipStreamSynchronize(c->stream));
    return RT_OK;
}

int rt_last_render_ms(rt_context *c, float *ms)
{
    if (!c || !ms) return fail(RT_ERR_INVALID, "null argument");
    RT_FLUSH_PENDING(c);
    if (!c->timed) return fail(RT_ERR_INVALID, "nothing rendered yet");
    RT_HIP_CHECK(hipSetDevice(c->device));
    RT_HIP_CHECK(hipEventSynchronize(c->ev_stop));
    RT_HIP_CHECK(hipEventElapsedTime(ms, c->ev_start, c->ev_stop));
    return RT_OK;
}

int rt_last_render_frames(rt_context *c, uint32_t *frames)
{
    if (!c || !frames) return fail(RT_ERR_INVALID, "null argument");
    RT_FLUSH_PENDING(c);
    if (!c->timed) return fail(RT_ERR_INVALID, "nothing rendered yet");
    *frames = c->timed_frames;
    return RT_OK;
}

int rt_read_accum(rt_context *c, float *rgb)
{
    if (!c || !rgb) return fail(RT_ERR_INVALID, "null argument");
    RT_FLUSH_PENDING(c);
    RT_HIP_CHECK(hipSetDevice(c->device));
    size_t px = (size_t)c->width * c->height;
    std::vector<float> planes(px * 3);
    RT_HIP_CHECK(hipMemcpyAsync(planes.data(), c->fb.accum, px * 3 * sizeof(float), hipMemcpyDeviceToHost, c->stream));
    RT_HIP_CHECK(hipStreamSynchronize(c->stream));
    for (size_t i = 0; i < px; i++) {
        rgb[3 * i + 0] = planes[i];
        rgb[3 * i + 1] = planes[px + i];
        rgb[3 * i + 2] = planes[2 * px + i];
    }
    return RT_OK;
}

int rt_write_accum(rt_context *c, const float *rgb)
{
    if (!c || !rgb) return fail(RT_ERR_INVALID, "null argument");
    RT_FLUSH_PENDING(c);
    RT_HIP_CHECK(hipSetDevice(c->device));
    size_t px = (size_t)c->width * c->height;
    std::vector<float> planes(px * 3);
    for (size_t i = 0; i < px; i++) {
        planes[i] = rgb[3 * i + 0];
        planes[px + i] = rgb[3 * i + 1];
        planes[2 * px + i] = rgb[3 * i + 2];
    }
    RT_HIP_CHECK(hipMemcpyAsync(c->fb.accum, planes.data(), px * 3 * sizeof(float), hipMemcpyHostToDevice, c->stream));
    RT_HIP_CHECK(hipStreamSynchronize(c->stream)); // `planes` goes out of scope
    c->launch_serial++; // the planes changed: the next rt_present makes a new image
    return RT_OK;
}

int rt_present(rt_context *c, uint8_t *rgba8, int wait, uint32_t *frame_index)
{
    if (!c || !rgba8) return fail(RT_ERR_INVALID, "null argument");
    if (wait) RT_FLUSH_PENDING(c);
    RT_HIP_CHECK(hipSetDevice(c->device));
    const size_t px = (size_t)c->width * c->height;
    if (!c->present_done[1]) {
        // made on first use; a call that failed half way is completed by the next one (present_done[1] is made last)
        if (!c->copy_stream) RT_HIP_CHECK(hipStreamCreateWithFlags(&c->copy_stream, hipStreamNonBlocking));
        for (int k = 0; k < 2; k++) {
            if (!c->d_present[k]) RT_HIP_CHECK(hipMalloc((void **)&c->d_present[k], px * 4));
            if (!c->h_present[k]) RT_HIP_CHECK(hipHostMalloc((void **)&c->h_present[k], px * 4, hipHostMallocDefault));
            if (!c->present_ready[k]) RT_HIP_CHECK(hipEventCreateWithFlags(&c->present_ready[k], hipEventDisableTiming));
            if (!c->present_done[k]) RT_HIP_CHECK(hipEventCreateWithFlags(&c->present_done[k], hipEventDisableTiming));
        }
    }
    // newest image asked for so far
    int newest = -1;
    for (int k = 0; k < 2; k++)
        if (c->present_used[k] && (newest < 0 || c->present_serial[k] > c->present_serial[newest])) newest = k;
    if (newest < 0 || c->present_serial[newest] != c->launch_serial) {
        // the planes have changed since: convert on the context's stream, behind the frames launched so far, and copy
        // to the host on the copy stream.  The slot taken is the older one; its previous copy is over by now, or waited for.
        int k = c->present_next;
        c->present_next ^= 1;
        if (c->present_used[k]) RT_HIP_CHECK(hipEventSynchronize(c->present_done[k]));
        hipLaunchKernelGGL(present_rgba8_kernel, dim3((unsigned)((px + 255) / 256)), dim3(256), 0, c->stream, c->fb.accum, c->d_present[k], px);
        RT_HIP_CHECK(hipGetLastError());
        RT_HIP_CHECK(hipEventRecord(c->present_ready[k], c->stream));
        RT_HIP_CHECK(hipStreamWaitEvent(c->copy_stream, c->present_ready[k], 0));
        RT_HIP_CHECK(hipMemcpyAsync(c->h_present[k], c->d_present[k], px * 4, hipMemcpyDeviceToHost, c->copy_stream));
        RT_HIP_CHECK(hipEventRecord(c->present_done[k], c->copy_stream));
        // (the next convert into this slot cannot overtake this copy: the host waits for present_done[k] above first)
        c->present_used[k] = true;
        c->present_serial[k] = c->launch_serial;
        c->present_frame[k] = c->launched_frame;
        newest = k;
    }
    int from = -1;
    if (wait) {
        RT_HIP_CHECK(hipEventSynchronize(c->present_done[newest]));
        from = newest;
    } else {
        // the newest image whose copy has arrived, without waiting
        for (int k = 0; k < 2; k++)
            if (c->present_used[k] && hipEventQuery(c->present_done[k]) == hipSuccess && (from < 0 || c->present_serial[k] > c->present_serial[from])) from = k;
    }
    if (from >= 0) std::memcpy(rgba8, c->h_present[from], px * 4);
    if (frame_index) *frame_index = from >= 0 ? c->present_frame[from] : 0u;
    return RT_OK;
}

int rt_read_frame_f16(rt_context *c, uint16_t *rgba)
{
    if (!c || !rgba) return fail(RT_ERR_INVALID, "null argument");
    RT_FLUSH_PENDING(c);
    RT_HIP_CHECK(hipSetDevice(c->device));
    size_t px = (size_t)c->width * c->height;
    RT_HIP_CHECK(hipMemcpyAsync(rgba, c->fb.frame_f16, px * 4 * sizeof(uint16_t), hipMemcpyDeviceToHost, c->stream));
    RT_HIP_CHECK(hipStreamSynchronize(c->stream));
    return RT_OK;
}

int rt_read_primary_ids(rt_context *c, int32_t *ids)
{
    if (!c || !ids) return fail(RT_ERR_INVALID, "null argument");
    RT_FLUSH_PENDING(c);
    RT_HIP_CHECK(hipSetDevice(c->device));
    size_t px = (size_t)c->width * c->height;
    RT_HIP_CHECK(hipMemcpyAsync(ids, c->fb.ids, px * sizeof(int32_t), hipMemcpyDeviceToHost, c->stream));
    RT_HIP_CHECK(hipStreamSynchronize(c->stream));
    return RT_OK;
}

int rt_set_counting(rt_context *c, int enable)
{
    if (!c) return fail(RT_ERR_INVALID, "null context");
    RT_FLUSH_PENDING(c);
    c->counting = enable != 0;
    return RT_OK;
}

// Undocumented diagnostic: wave-cycle totals per phase of the instrumented persistent kernel
// (inner, triangle, shadow hand-over, shade, whole kernel) in out[0..4]; out[8] traversal turns and
// out[9..14] lanes per state summed over them (inner node, leaf, shadow done, ready to shade, idle,
// dead); valid after a counting run.
int rt_debug_phase_cycles(rt_context *c, uint64_t out[16])
{
    if (!c || !out) return fail(RT_ERR_INVALID, "null argument");
    RT_FLUSH_PENDING(c);
    RT_HIP_CHECK(hipSetDevice(c->device));
    unsigned long long tmp[16];
    RT_HIP_CHECK(hipMemcpyAsync(tmp, c->fb.counters + 16, sizeof tmp, hipMemcpyDeviceToHost, c->stream));
    RT_HIP_CHECK(hipStreamSynchronize(c->stream));
    for (int i = 0; i < 16; i++) out[i] = tmp[i];
    return RT_OK;
}

int rt_debug_wave_trace(rt_context *c, uint64_t *out, size_t max_values, size_t *n_values)
{
    if (!c || !n_values) return fail(RT_ERR_INVALID, "null argument");
    RT_FLUSH_PENDING(c);
    RT_HIP_CHECK(hipSetDevice(c->device));
    *n_values = c->wave_trace_used;
    if (!out || max_values == 0 || c->wave_trace_used == 0) return RT_OK;
    size_t n = std::min(max_values, c->wave_trace_used);
    RT_HIP_CHECK(hipStreamSynchronize(c->stream));
    for (hipStream_t s2 : c->aux)
        if (s2) RT_HIP_CHECK(hipStreamSynchronize(s2));
    RT_HIP_CHECK(hipMemcpy(out, c->fb.wave_trace, n * sizeof(uint64_t), hipMemcpyDeviceToHost));
    return RT_OK;
}

int rt_read_counters(rt_context *c, uint64_t counters[16])
{
    if (!c || !counters) return fail(RT_ERR_INVALID, "null argument");
    RT_FLUSH_PENDING(c);
    RT_HIP_CHECK(hipSetDevice(c->device));
    unsigned long long tmp[16];
    RT_HIP_CHECK(hipMemcpyAsync(tmp, c->fb.counters, sizeof tmp, hipMemcpyDeviceToHost, c->stream));
    RT_HIP_CHECK(hipStreamSynchronize(c->stream));
    for (int i = 0; i < 16; i++) counters[i] = tmp[i];
    return RT_OK;
}

int rt_take_triangle_tests(rt_context *c, uint32_t *triangle_tests)
{
    if (!c || !triangle_tests) return fail(RT_ERR_INVALID, "null argument");
    uint64_t cnt[16];
    int rc = rt_read_counters(c, cnt);
    if (rc) return rc;
    uint64_t rays = cnt[0] - c->tests_taken_rays;
    c->tests_taken_rays = cnt[0];
    *triangle_tests = (uint32_t)(rays * (uint64_t)c->view.n_tris); // u32 wrap of raytracer_compute.wgsl:27
    return RT_OK;
}

int rt_owned_rows(rt_context *c, int *rows)
{
    if (!c || !rows) return fail(RT_ERR_INVALID, "null argument");
    *rows = c->local_rows;
    return RT_OK;
}

int rt_export_owned_rows(rt_context *c, void *dst_device)
{
    if (!c || !dst_device) return fail(RT_ERR_INVALID, "null argument");
    RT_FLUSH_PENDING(c);
    RT_HIP_CHECK(hipSetDevice(c->device));
    size_t n = (size_t)c->local_rows * c->width;
    if (n == 0) return RT_OK;
    hipLaunchKernelGGL(export_rows_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, c->stream, c->fb.accum,
                       (float *)dst_device, c->width, c->height, c->local_rows, c->sm);
    RT_HIP_CHECK(hipGetLastError());
    return RT_OK;
}

int rt_import_rows(rt_context *c, const void *src_device, int from_index)
{
    if (!c || !src_device) return fail(RT_ERR_INVALID, "null argument");
    RT_FLUSH_PENDING(c);
    if (from_index < 0 || from_index >= c->sm.count) return fail(RT_ERR_INVALID, "from_index out of range");
    RT_HIP_CHECK(hipSetDevice(c->device));
    StripeMap sm = c->sm;
    sm.index = from_index;
    int rows = owned_rows_of(c->height, sm);
    size_t n = (size_t)rows * c->width;
    if (n == 0) return RT_OK;
    hipLaunchKernelGGL(import_rows_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, c->stream, c->fb.accum,
                       (const float *)src_device, c->width, c->height, rows, sm, n);
    RT_HIP_CHECK(hipGetLastError());
    c->launch_serial++; // the planes changed: the next rt_present makes a new image
    return RT_OK;
}

int rt_import_gathered_rows(rt_context *c, const void *src_device, int rows_per_block, int skip_index)
{
    if (!c || !src_device) return fail(RT_ERR_INVALID, "null argument");
    RT_FLUSH_PENDING(c);
    RT_HIP_CHECK(hipSetDevice(c->device));
    for (int from = 0; from < c->sm.count; from++) { // nothing is written unless every block fits
        StripeMap sm = c->sm;
        sm.index = from;
        if (from != skip_index && owned_rows_of(c->height, sm) > rows_per_block)
            return fail(RT_ERR_INVALID, "rows_per_block is smaller than a stripe's block");
    }
    for (int from = 0; from < c->sm.count; from++) {
        if (from == skip_index) continue;
        StripeMap sm = c->sm;
        sm.index = from;
        int rows = owned_rows_of(c->height, sm);
        size_t n = (size_t)rows * c->width, block = (size_t)3 * rows_per_block * c->width;
        if (n == 0) continue;
        hipLaunchKernelGGL(import_rows_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, c->stream, c->fb.accum,
                           (const float *)src_device + (size_t)from * block, c->width, c->height, rows, sm,
                           (size_t)rows_per_block * c->width);
        RT_HIP_CHECK(hipGetLastError());
    }
    c->launch_serial++; // the planes changed: the next rt_present makes a new image
    return RT_OK;
}

int rt_rows_of_stripe(rt_context *c, int stripe_index, int *rows)
{
    if (!c || !rows) return fail(RT_ERR_INVALID, "null argument");
    if (stripe_index < 0 || stripe_index >= c->sm.count) return fail(RT_ERR_INVALID, "stripe_index out of range");
    StripeMap sm = c->sm;
    sm.index = stripe_index;
    *rows = owned_rows_of(c->height, sm);
    return RT_OK;
}

int rt_read_owned_rows(rt_context *c, float *planes)
{
    if (!c || !planes) return fail(RT_ERR_INVALID, "null argument");
    RT_FLUSH_PENDING(c);
    RT_HIP_CHECK(hipSetDevice(c->device));
    size_t n = (size_t)c->local_rows * c->width;
    if (n == 0) return RT_OK;
    int rc = ensure_row_staging(c, n * 3 * sizeof(float));
    if (rc) return rc;
    float *tmp = c->d_row_staging;
    hipLaunchKernelGGL(export_rows_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, c->stream, c->fb.accum, tmp, c->width,
                       c->height, c->local_rows, c->sm);
    RT_HIP_CHECK(hipGetLastError());
    RT_HIP_CHECK(hipMemcpyAsync(planes, tmp, n * 3 * sizeof(float), hipMemcpyDeviceToHost, c->stream));
    RT_HIP_CHECK(hipStreamSynchronize(c->stream));
    return RT_OK;
}

int rt_write_rows(rt_context *c, const float *planes, int from_index)
{
    if (!c || !planes) return fail(RT_ERR_INVALID, "null argument");
    if (from_index < 0 || from_index >= c->sm.count) return fail(RT_ERR_INVALID, "from_index out of range");
    RT_FLUSH_PENDING(c);
    RT_HIP_CHECK(hipSetDevice(c->device));
    StripeMap sm = c->sm;
    sm.index = from_index;
    size_t n = (size_t)owned_rows_of(c->height, sm) * c->width;
    if (n == 0) return RT_OK;
    int rc = ensure_row_staging(c, n * 3 * sizeof(float));
    if (rc) return rc;
    float *tmp = c->d_row_staging;
    RT_HIP_CHECK(hipMemcpyAsync(tmp, planes, n * 3 * sizeof(float), hipMemcpyHostToDevice, c->stream));
    if ((rc = rt_import_rows(c, tmp, from_index))) return rc;
    RT_HIP_CHECK(hipStreamSynchronize(c->stream)); // `planes` may be reused by the caller, and the staging block by the next call
    return RT_OK;
}

int rt_frames_per_launch(rt_context *c, uint32_t *frames)
{
    if (!c || !frames) return fail(RT_ERR_INVALID, "null argument");
    *frames = frames_per_launch(c);
    return RT_OK;
}

int rt_scene_stats(rt_context *c, uint64_t out[4])
{
    if (!c || !out) return fail(RT_ERR_INVALID, "null argument");
    if (!c->has_scene) return fail(RT_ERR_INVALID, "rt_set_scene has not been called");
    out[0] = c->host_scene.nodes.size() / 4;
    out[1] = c->host_scene.n_leaves;
    out[2] = c->host_scene.max_depth;
    out[3] = c->scene_bytes;
    return RT_OK;
}

} // extern "C"
