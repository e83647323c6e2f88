// A run of stride-1 3x3 convolutions of one shape (the 2 k convolutions of k consecutive BasicBlocks, networks/backbone.py:136-159)
// as ONE launch of the Winograd kernel's block body (conv_wino_body.inc), layer after layer, with a per-region dataflow wait in
// place of the launch boundary.  EXPERIMENTAL (SMOS_WINO_CHAIN=1; off by default): what it is for, what it measured and why it
// is not the default are in DESIGN.md section 4.10.
//
//   grid       = the items of one layer (region x cout tile), one item per block and layer, every block resident (the host checks
//                n_items <= CUs x resident blocks): block (region r, cout tile ct) computes item (r, ct) of layer 0, 1, .. in turn.
//   dependency = layer L reads layer L - 1's output in region r and its 8 neighbours (the 3x3 halo), all cout tiles.  done[L][r]
//                counts the finished items of region r (nct when complete; the counters are monotonic across launches: a launch
//                waits for `target` = launch number x nct, so nothing is cleared in between).
//   coherence  = the XCDs' L2s are not coherent with each other: every activation load and output store of the body carries the
//                agent-scope bit (sc1: WINO_COH), so the data itself bypasses the non-coherent levels; no cache-wide fence.
//   publish    = every wave waits for its stores (vmcnt(0)); block barrier; one relaxed agent-scope increment.
//   consume    = nine threads poll the neighbours' counters (agent-scope loads, s_sleep between polls, bounded: past 2^21 polls
//                the block raises *err and goes on -- wrong results, never a hang); block barrier.
//   residual   = a layer's residual input is an EARLIER layer's output in the block's own region: complete by transitivity.
//   buffers    = every layer writes its own output map (the engine allocates them as it does launch by launch): no reuse, no
//                write-after-read hazard.
#include <stdlib.h>

#include "conv_wino_common.h"

namespace smos {

constexpr int kChainMax = 12;

struct ChainArgs {
  const float* x[kChainMax];
  const float4* w[kChainMax];
  const float* bias[kChainMax];
  const float* res[kChainMax];
  float* out[kChainMax];
  int64_t xp[kChainMax], rp[kChainMax], op[kChainMax];
  float slope[kChainMax];
  float* sums;          // channel sums of the LAST layer (conv_wino<.., SUMS>), or null
  int* done;            // [n_layers][B * yb * xb]
  int* err;
  int n_layers, target;
  int B, H, W, nchunk, nct, yb, xb, n_items, cout;
  int x_bytes[kChainMax], r_bytes[kChainMax], o_bytes[kChainMax];
};

// agent scope (sc1) on every activation load and output store of the body: a layer's input was written during this launch by
// blocks on other XCDs, whose L2s are not coherent with this one's; with the accesses themselves coherent no cache-wide
// write-back / invalidate is needed at the layer boundary (the release / acquire FENCES of the first version -- buffer_wbl2 and
// buffer_inv once per wave and layer -- cost more than the launch boundaries they replaced: 52.9 vs 33.7 us per layer)
#define WINO_COH 16
template <int MB, bool RES, bool SUMS>
__device__ __forceinline__ void chain_layer(const WinoArgs& a, float* lds) {
#include "conv_wino_body.inc"
}

template <int MB>
__global__ __launch_bounds__(256, 2) void conv_wino_chain(ChainArgs c) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  // the block's item, as the body's group order assigns it with one item per block: logical block = XCD-contiguous index,
  // region = lblock / nct
  const int nb = (int)gridDim.x, xq = nb >> 3, xr = nb & 7, xcd = (int)blockIdx.x & 7;
  const int lblock = (xcd < xr ? xcd * (xq + 1) : xr * (xq + 1) + (xcd - xr) * xq) + ((int)blockIdx.x >> 3);
  const int region = lblock / c.nct, regions = c.n_items / c.nct;
  for (int L = 0; L < c.n_layers; ++L) {
    if (L > 0) {
      const int rx = region % c.xb, ry = (region / c.xb) % c.yb, rb = region / (c.xb * c.yb);
      if ((int)threadIdx.x < 9) {
        const int tid = threadIdx.x;
        const int yy = ry + tid / 3 - 1, xx = rx + tid % 3 - 1;
        if (yy >= 0 && yy < c.yb && xx >= 0 && xx < c.xb) {
          const int* p = c.done + (L - 1) * regions + (rb * c.yb + yy) * c.xb + xx;
          int polls = 0;
          while (__hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < c.target) {
            __builtin_amdgcn_s_sleep(8);
            if (++polls > (1 << 21)) {                  // never hang: raise the flag and go on with whatever is there
              __hip_atomic_store(c.err, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
              break;
            }
          }
        }
      }
      __syncthreads();        // (also a compiler barrier: the layer's loads stay behind the wait)
    }
    WinoArgs a;
    a.x = c.x[L]; a.w = c.w[L]; a.bias = c.bias[L]; a.res = c.res[L]; a.out = c.out[L];
    a.sums = (L == c.n_layers - 1) ? c.sums : nullptr;
    a.xp = c.xp[L]; a.rp = c.rp[L]; a.op = c.op[L];
    a.B = c.B; a.H = c.H; a.W = c.W; a.nchunk = c.nchunk; a.nct = c.nct; a.yb = c.yb; a.xb = c.xb; a.n_items = c.n_items;
    a.slope = c.slope[L];
    a.x_bytes = c.x_bytes[L]; a.r_bytes = c.r_bytes[L]; a.o_bytes = c.o_bytes[L]; a.cout = c.cout;
    a.group = 1;
    if (a.res) chain_layer<MB, true, false>(a, lds);
    else if (a.sums) chain_layer<MB, false, true>(a, lds);
    else chain_layer<MB, false, false>(a, lds);
    if (L + 1 < c.n_layers) {
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");     // this wave's (write-through) stores have been acknowledged
      __syncthreads();
      if (threadIdx.x == 0) __hip_atomic_fetch_add(c.done + L * regions + region, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
  }
}

}  // namespace smos

using namespace smos;

extern "C" int64_t smos_conv_wino_chain_ws_ints(int64_t n_layers, int64_t B, int64_t H, int64_t W) {
  return n_layers * B * ((H + 7) / 8) * ((W + 31) / 32) + 1;
}

// res_from[L]: -1 no residual; 0 the chain's input x0; j > 0 the output of layer j - 1.  acts[L]: 0 none, 1 ReLU, 2 LeakyReLU.
// chan_sums: channel sums of the last layer's output (as smos_conv_wino_cl's; that layer then takes no residual), or NULL.
// ws: smos_conv_wino_chain_ws_ints int32 words, zeroed ONCE by the caller; launch_no = 1, 2, 3, .. counts the launches that used
// this ws with these sizes (the completion counters are monotonic).  ws[last] is raised when a wait gave up (results invalid).
extern "C" int smos_conv_wino_chain_cl(int32_t n_layers, const float* x0, int64_t x0_pitch, const float* const* wprep,
                                       const float* const* bias, const int32_t* res_from, float* const* outs, const int64_t* out_pitches,
                                       const int32_t* acts, float* chan_sums, int32_t* ws, int32_t launch_no, int64_t B, int64_t H,
                                       int64_t W, int64_t C, int32_t mb, smos_stream_t stream) {
  SMOS_REQUIRE(n_layers >= 2 && n_layers <= kChainMax && x0 && wprep && bias && res_from && outs && out_pitches && acts && ws &&
                   launch_no >= 1, "conv_wino_chain_cl: 2 .. 12 layers, no null argument, launch_no >= 1");
  SMOS_REQUIRE(B > 0 && H > 0 && W > 0 && C > 0 && (mb == 1 || mb == 2) && C % (16 * mb) == 0 && C <= 2048 && x0_pitch >= C &&
                   x0_pitch % 4 == 0, "conv_wino_chain_cl: C must be a multiple of 16 * mb (mb in {1, 2})");
  const int64_t yb = (H + 7) / 8, xb = (W + 31) / 32, nct = C / (16 * mb), regions = B * yb * xb, n_items = regions * nct;
  SMOS_REQUIRE((int64_t)launch_no * nct < (1LL << 30), "conv_wino_chain_cl: launch counter exhausted (use a fresh ws)");
  ChainArgs c;
  for (int L = 0; L < n_layers; ++L) {
    const float* xin = L == 0 ? x0 : outs[L - 1];
    const int64_t xpitch = L == 0 ? x0_pitch : out_pitches[L - 1];
    SMOS_REQUIRE(wprep[L] && outs[L] && out_pitches[L] >= C && out_pitches[L] % 4 == 0 && acts[L] >= 0 && acts[L] <= 2 &&
                     res_from[L] >= -1 && res_from[L] <= L, "conv_wino_chain_cl: bad layer description");
    SMOS_REQUIRE(B * H * W * xpitch * 4 < (1LL << 31) && B * H * W * out_pitches[L] * 4 < (1LL << 31),
                 "conv_wino_chain_cl: a tensor larger than 2 GiB (32-bit buffer offsets)");
    for (int j = 0; j < L; ++j) SMOS_REQUIRE(outs[j] != outs[L], "conv_wino_chain_cl: every layer needs its own output map");
    SMOS_REQUIRE(outs[L] != x0, "conv_wino_chain_cl: a layer may not write the chain's input");
    const float* r = nullptr;
    int64_t rp = 0;
    if (res_from[L] == 0) { r = x0; rp = x0_pitch; }
    if (res_from[L] > 0) { r = outs[res_from[L] - 1]; rp = out_pitches[res_from[L] - 1]; }
    SMOS_REQUIRE(!(r && L == n_layers - 1 && chan_sums), "conv_wino_chain_cl: channel sums need a last layer without residual");
    SMOS_REQUIRE(((reinterpret_cast<uintptr_t>(xin) | reinterpret_cast<uintptr_t>(outs[L]) | reinterpret_cast<uintptr_t>(r) |
                   reinterpret_cast<uintptr_t>(bias[L]) | reinterpret_cast<uintptr_t>(wprep[L])) & 15) == 0,
                 "conv_wino_chain_cl: pointers must be 16-byte aligned");
    c.x[L] = xin; c.w[L] = reinterpret_cast<const float4*>(wprep[L]); c.bias[L] = bias[L]; c.res[L] = r; c.out[L] = outs[L];
    c.xp[L] = xpitch; c.rp[L] = rp; c.op[L] = out_pitches[L];
    c.slope[L] = acts[L] == 0 ? 1.0f : acts[L] == 1 ? 0.0f : 0.01f;
    c.x_bytes[L] = (int)(B * H * W * xpitch * 4);
    c.r_bytes[L] = r ? (int)(B * H * W * rp * 4) : 0;
    c.o_bytes[L] = (int)(B * H * W * out_pitches[L] * 4);
  }
  for (int L = n_layers; L < kChainMax; ++L) {
    c.x[L] = nullptr; c.w[L] = nullptr; c.bias[L] = nullptr; c.res[L] = nullptr; c.out[L] = nullptr;
    c.xp[L] = c.rp[L] = c.op[L] = 0; c.slope[L] = 1.0f; c.x_bytes[L] = c.r_bytes[L] = c.o_bytes[L] = 0;
  }
  c.sums = chan_sums;
  c.done = ws;
  c.err = ws + n_layers * regions;
  c.n_layers = n_layers;
  c.target = launch_no * (int)nct;
  c.B = (int)B; c.H = (int)H; c.W = (int)W; c.nchunk = (int)(C / 16); c.nct = (int)nct; c.yb = (int)yb; c.xb = (int)xb;
  c.n_items = (int)n_items; c.cout = (int)C;
  const size_t lds = (size_t)2 * kWInWords * sizeof(float) + (size_t)3 * 256 * mb * sizeof(float4);
  KernelSetup ks;
  const void* fn = mb == 1 ? reinterpret_cast<const void*>(&conv_wino_chain<1>) : reinterpret_cast<const void*>(&conv_wino_chain<2>);
  if (int rc = kernel_setup(fn, lds, 256, &ks, "conv_wino_chain_cl")) return rc;
  // every block must be able to be resident at once: a block waits for its neighbours' previous layer
  SMOS_REQUIRE(ks.per_cu >= 1 && n_items <= (int64_t)ks.cus * (ks.per_cu < 2 ? ks.per_cu : 2),
               "conv_wino_chain_cl: more work items than resident blocks (use the launch-by-launch form)");
  if (mb == 1) hipLaunchKernelGGL(conv_wino_chain<1>, dim3((unsigned)n_items), dim3(256), lds, (hipStream_t)stream, c);
  else hipLaunchKernelGGL(conv_wino_chain<2>, dim3((unsigned)n_items), dim3(256), lds, (hipStream_t)stream, c);
  return check_launch("conv_wino_chain_cl");
}
