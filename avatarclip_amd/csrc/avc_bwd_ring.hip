// Role-specialised backward of the point MLP (main.py:537; fields.py:96-107 double backward): ONE persistent launch in which,
// per XCD, PRODUCER workgroups run the three backward sweeps (csrc/avc_bwd_body.h, exactly the work of avc_render_points_bwd)
// and CONSUMER workgroups own the fp32 accumulators of one weight-gradient product each and contract it as the producers go.
// The gradient-type operand of such a product -- abar of a middle SDF layer, 8 tiles per 32-point block: a pure hand-off tile,
// written by the backward sweep and read only by the product abar_m (x) h_in -- does not travel through the G region in HBM:
// the producer writes it into a slot of a small ring that lives in the XCD's write-back L2, a consumer of the SAME XCD copies it
// from there into LDS (global -> LDS DMA with device scope, i.e. served by that L2) next to the forward-type operand h_in, which
// it streams from the F region exactly like avc_weight_grad_all.  Every other product keeps going through the panels
// (avc_weight_grad_all on the remaining pairs).
//
// Hand-off unit = one producer workgroup iteration of one layer: 8 wavefronts x HT tiles x 2 KiB (128 KiB for the full nets).
// Protocol (same-XCD, no L2 write-back; measured word by word in scripts/ubench/ubench3.hip, profiles/r03_ubench3.txt):
//   producer  thread 0: q = ticket of (XCD, product) ; wait until slot q % NS is free ; meta[slot] = first block ; barrier ;
//             every wavefront: plain stores of its tiles -> s_waitcnt vmcnt(0) (they are in the XCD's L2) -> arrive[slot] += 1
//   consumer  (takes the tickets q = k, k + C, ..): poll arrive[slot] == 8 (q / NS + 1) ; copy + contract the unit's blocks with
//             the LDS ring of avc_wgrad.hip ; after the barrier of the last block: freed[slot] = q + 1
// Roles are dealt per XCD by the hardware XCC id and an arrival ticket (the first ntypes * cpt workgroups of an XCD become
// consumers), work is claimed dynamically from one chip-wide counter, so nothing depends on the block -> XCD map or on how many
// workgroups an XCD receives: an XCD without consumers would stall its producers, which is why consumers take the FIRST tickets.
// What DOES rely on placement is the payload path: plain stores + device-scope loads are coherent only through a shared L2,
// i.e. between workgroups that read the same XCC id.  Every spin is bounded; a time-out raises the abort word, every role leaves,
// and the host turns a non-zero error word into NaN gradients (engine.py).
#include "avc_bwd_body.h"
#include "avc_wgrad_body.h"
#include "../../include/avc.h"

#define RING_NTY_MAX 8
#define RING_NS_MAX 16
#define RC_LINE 32                                  // u32 words per control line (128 B)
#define RC_NEXT 0                                   // next 256-point group (chip-wide)
#define RC_ABORT (1 * RC_LINE)
#define RC_ERR (2 * RC_LINE)                        // [0] spin time-outs, [1] first failing site
#define RC_STAT (3 * RC_LINE)                       // u64 counters (10-ns ticks / counts), see avc.h
#define RC_XCD (5 * RC_LINE)                        // per XCD: lines ticket, started, finished, head[RING_NTY_MAX]
#define RC_XCD_LINES (3 + RING_NTY_MAX)
#define RC_SLOT0 (RC_XCD + 8 * RC_XCD_LINES * RC_LINE)   // per (xcd, type, slot): one line {arrive, freed, meta}
#define RC_WORDS (RC_SLOT0 + 8 * RING_NTY_MAX * RING_NS_MAX * RC_LINE)
#ifndef RING_SPIN_LIMIT
#define RING_SPIN_LIMIT (1 << 21)                   // polls (~0.3 us each with the sleep): ~0.6 s
#endif
// timing / counter experiments only (the products are garbage): 1 = the consumer takes its forward-type operand from the ring slot too
// (the abar tiles a second time) instead of from the F region in HBM -- a consumer that never waits for HBM; 2 = the producer
// additionally writes a second copy of its tiles into the slot (what handing BOTH operands through the ring would cost it)
#ifndef RING_EXP_B
#define RING_EXP_B 0
#endif
#define RING_UNIT_TILES(HT) ((RING_EXP_B == 2 ? 2 : 1) * (HT))   // tiles per wavefront and hand-off unit
#ifndef RING_DMA_AUX
#define RING_DMA_AUX 16                             // sc1: device scope -- the copy is served by the XCD's L2, never by this CU's L1
#endif

struct RingParams {
  unsigned* ctl;          // RC_WORDS u32, zeroed by the launcher
  char* ring;             // payload [xcd][type][slot][8 waves][HT tiles][2048 B]
  float* partial;         // [type][8 * cpt][HT * HT * 1024]
  float* bias_partial;    // [type][8 * cpt][HT * 32]
  int ntypes, cpt, nslots;
  unsigned ngroups;
  int pb[RING_NTY_MAX];   // F-region tile index of the forward-type operand of product `type`
};

__device__ __forceinline__ int xcc_id() { return __builtin_amdgcn_s_getreg((3 << 11) | (0 << 6) | 20) & 0x7; }
__device__ __forceinline__ unsigned ld_agent(const unsigned* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ void st_agent(unsigned* p, unsigned v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ unsigned add_agent(unsigned* p, unsigned v) {
  const unsigned r = __hip_atomic_fetch_add(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // performed before anything this wave does next
  return r;
}
__device__ __forceinline__ unsigned* xcd_line(const RingParams& rp, int xcd, int which) { return rp.ctl + RC_XCD + (xcd * RC_XCD_LINES + which) * RC_LINE; }
__device__ __forceinline__ unsigned* slot_line(const RingParams& rp, int xcd, int type, unsigned slot) {
  return rp.ctl + RC_SLOT0 + ((xcd * RING_NTY_MAX + type) * RING_NS_MAX + slot) * RC_LINE;
}
__device__ __forceinline__ void stat_add(const RingParams& rp, int which, unsigned long long v) {
  atomicAdd(reinterpret_cast<unsigned long long*>(rp.ctl + RC_STAT) + which, v);
}
__device__ __forceinline__ void raise_abort(const RingParams& rp, unsigned site) {
  atomicAdd(rp.ctl + RC_ERR, 1u);
  atomicCAS(rp.ctl + RC_ERR + 1, 0u, site);
  st_agent(rp.ctl + RC_ABORT, 1u);
}

// ---------------------------------------------------------------------------------------------------------------- producer side
template <class N>
struct RingProd {
  static constexpr bool on = true;
  const RingParams& rp;
  int xcd;
  unsigned* s_w;                      // LDS: [0] slot of the current hand-off, [1] abort seen
  unsigned long long t_alloc, t_hand; // thread 0: ticks spent allocating (ticket + waiting for a free slot) / in the whole hand-off
  unsigned polls;

  template <class N2>
  __device__ __forceinline__ void handoff(int type, const b8 (&f)[N2::HK], long blk0, int lane, int wv) {
    unsigned long long t0 = 0;
    if (threadIdx.x == 0) {
      t0 = wall_clock64();
      const unsigned ns = (unsigned)rp.nslots;
      const unsigned q = add_agent(xcd_line(rp, xcd, 3 + type), 1u);
      const unsigned slot = q % ns;
      unsigned* line = slot_line(rp, xcd, type, slot);
      const unsigned need = q >= ns ? q - ns + 1 : 0u;
      unsigned spins = 0;
      bool bad = false;
      while (ld_agent(line + 1) < need) {
        __builtin_amdgcn_s_sleep(4);
        ++spins;
        if ((spins & 63u) == 0 && ld_agent(rp.ctl + RC_ABORT)) { bad = true; break; }
        if (spins > RING_SPIN_LIMIT) { raise_abort(rp, 1u); bad = true; break; }
      }
      polls += spins;
      st_agent(line + 2, (unsigned)blk0);            // meta: the unit's first block (ordered before this wave's arrival below)
      s_w[0] = slot;
      if (bad) s_w[1] = 1u;
      t_alloc += wall_clock64() - t0;
    }
    __syncthreads();
    const unsigned slot = __builtin_amdgcn_readfirstlane(s_w[0]);
    char* dst = rp.ring + ((((long)xcd * rp.ntypes + type) * rp.nslots + slot) * BWD_WPB + wv) * (long)(RING_UNIT_TILES(N2::HT) * 2048);
    const PanelPtr pp = panel_ptr(dst, lane);
    tiles_store<true, N2::HT>(pp, 0, f);             // plain stores: the lines stay in this XCD's write-back L2
    if (RING_EXP_B == 2) tiles_store<true, N2::HT>(pp, N2::HT, f);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); // ... and are there now
    if (lane == 0) __hip_atomic_fetch_add(slot_line(rp, xcd, type, slot), 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (threadIdx.x == 0) t_hand += wall_clock64() - t0;
  }
};

// ---------------------------------------------------------------------------------------------------------------- consumer side
// product `type`: A = the ring's abar tiles (bf16, HT tiles per block), B = HT forward-type tiles (f16) at F-region tile pb.
// Wave layout of avc_wgrad.hip's wide x wide case with the f16 operand on the B side: 2 x 4 waves, A tiles wa + 2 i (i < 4), B
// tiles wb + 4 k (k < 2).
template <class N>
__device__ __forceinline__ void ring_consumer(char* lds, const RingParams& rp, const BwdArgs& a, int xcd, int type, int k) {
  typedef PanelLayout<N> L;
  constexpr int NI = 4, NK = 2, WA = 2, WB = 4, TA = N::HT, TB = N::HT;
  constexpr int ntile = TA + TB, nchunk = ntile * 2, depth = WG_DEPTH, slot_bytes = WG_BUF_BYTES;
  static_assert(ntile <= WG_TILES_MAX, "LDS ring slot too small");
  const int lane = threadIdx.x & 63;
  const int wv = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
  const int wa = wv % WA, wb = wv / WA;
  const int my_chunks = (nchunk - wv + 7) >> 3;
  const int src_lo = ((lane >> 2) & 3) * 32 + 4 * (lane >> 4) + (lane & 3);
  const int lane_off = (((lane >> 5) * 2) * 16 + (((lane >> 4) & 1) * 2 + (lane & 1)) * 4 + ((lane & 15) >> 2)) * 16 + 8 * ((lane & 3) >> 1);
  const long nblk = (a.npts + 31) >> 5;
  const unsigned ns = (unsigned)rp.nslots;
  const char* const fbase = a.fpanels + (long)rp.pb[type] * 2048;
  const long fstride = (long)L::P_TILES * 2048;
  constexpr int UT = RING_UNIT_TILES(TA);
  const char* const rbase = rp.ring + (((long)xcd * rp.ntypes + type) * rp.nslots) * (long)(BWD_WPB * UT * 2048);
  facc acc[NI * NK];
#pragma unroll
  for (int q = 0; q < NI * NK; ++q)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[q][r] = 0.f;
  float bsum[NI];
#pragma unroll
  for (int i = 0; i < NI; ++i) bsum[i] = 0.f;
  unsigned long long t_wait = 0, units = 0;
  int ls = 0;   // LDS ring slot of the next block to contract
  for (unsigned q = (unsigned)k;; q += (unsigned)rp.cpt) {
    // ---- acquire unit q (every wavefront polls for itself: the answers are monotonic, so all eight agree)
    const unsigned slot = q % ns, need = BWD_WPB * (q / ns + 1);
    unsigned* line = slot_line(rp, xcd, type, slot);
    bool end = false;
    {
      const unsigned long long t0 = wall_clock64();
      unsigned spins = 0;
      while (ld_agent(line) < need) {
        if (ld_agent(rp.ctl + RC_NEXT) >= rp.ngroups) {   // all work is claimed: are this XCD's producers done, and is q beyond their last ticket?
          const unsigned fin = ld_agent(xcd_line(rp, xcd, 2));
          const unsigned sta = ld_agent(xcd_line(rp, xcd, 1));          // (read AFTER `finished`: fin == sta => nobody is in flight)
          if (fin == sta && ld_agent(xcd_line(rp, xcd, 3 + type)) <= q) { end = true; break; }
        }
        __builtin_amdgcn_s_sleep(4);
        ++spins;
        if ((spins & 63u) == 0 && ld_agent(rp.ctl + RC_ABORT)) { end = true; break; }
        if (spins > RING_SPIN_LIMIT) { raise_abort(rp, 2u); end = true; break; }
      }
      if (threadIdx.x == 0) t_wait += wall_clock64() - t0;
    }
    if (end) break;
    const long blk0 = (long)ld_agent(line + 2);
    int nv = (int)(nblk - blk0 < BWD_WPB ? nblk - blk0 : BWD_WPB);
    nv = __builtin_amdgcn_readfirstlane(nv);
    const char* const ra = rbase + (long)slot * (BWD_WPB * UT * 2048);
    auto issue = [&](int j, int s) {
      for (int c = wv; c < nchunk; c += 8) {
        const int tix = c >> 1;
        char* dst = lds + s * slot_bytes + c * 1024;
        if (tix < TA || RING_EXP_B != 0) {
          const int rt = tix < TA ? tix : (RING_EXP_B == 2 ? tix : tix - TA);
          const char* g = ra + ((long)j * UT + rt) * 2048 + (src_lo + (c & 1) * 16) * 16;
          __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)g, (__attribute__((address_space(3))) void*)dst, 16, 0, RING_DMA_AUX);
        } else {
          const char* g = fbase + (blk0 + j) * fstride + (long)(tix - TA) * 2048 + (src_lo + (c & 1) * 16) * 16;
          __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)g, (__attribute__((address_space(3))) void*)dst, 16, 0, WG_DMA_AUX);
        }
      }
    };
    for (int d = 0; d < depth - 1; ++d)
      if (d < nv) { int s = ls + d; if (s >= depth) s -= depth; issue(d, s); }
    for (int j = 0; j < nv; ++j) {
      int younger = nv - 1 - j;
      if (younger > depth - 2) younger = depth - 2;
      wait_vmcnt(younger * my_chunks);
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // my reads of the previous block are done
      __builtin_amdgcn_s_barrier();
      __builtin_amdgcn_sched_barrier(0);
      if (j + depth - 1 < nv) { int s = ls + depth - 1; if (s >= depth) s -= depth; issue(j + depth - 1, s); }
      lds_char* buf = (lds_char*)(lds + ls * slot_bytes) + lane_off;
      b8 av[NI][2];
#pragma unroll
      for (int i = 0; i < NI; ++i) {
        const int ta = wa + WA * i;
        if (ta < TA) { av[i][0] = tr_frag(buf + ta * 2048, 0); av[i][1] = tr_frag(buf + ta * 2048, 1); }
      }
      b8 bv[NK][2];
#pragma unroll
      for (int kk = 0; kk < NK; ++kk) {
        const int tb = wb + WB * kk;
        if (tb < TB) { bv[kk][0] = tr_frag(buf + (TA + tb) * 2048, 0); bv[kk][1] = tr_frag(buf + (TA + tb) * 2048, 1); }
      }
#pragma unroll
      for (int kk = 0; kk < NK; ++kk) { bv[kk][0] = f16_to_bf16(bv[kk][0]); bv[kk][1] = f16_to_bf16(bv[kk][1]); }
#pragma unroll
      for (int i = 0; i < NI; ++i) {
        if (wa + WA * i < TA) {
          if (wb == i) {   // the bias row sums of A tile wa + 2 i are the job of wave (wa, wb = i)
#pragma unroll
            for (int jj = 0; jj < 8; ++jj) bsum[i] += (float)av[i][0][jj] + (float)av[i][1][jj];
          }
#pragma unroll
          for (int kk = 0; kk < NK; ++kk) {
            if (wb + WB * kk < TB) {
              acc[i * NK + kk] = MF<b8>::mma(av[i][0], bv[kk][0], acc[i * NK + kk]);
              acc[i * NK + kk] = MF<b8>::mma(av[i][1], bv[kk][1], acc[i * NK + kk]);
            }
          }
        }
      }
      if (++ls == depth) ls = 0;
    }
    // past the barrier of the unit's last block: every wavefront's copies of the whole unit have landed -> the L2 slot is free
    if (threadIdx.x == 0) { st_agent(line + 1, q + 1); ++units; }
  }
  const int row = xcd * rp.cpt + k, nrow = 8 * rp.cpt;
  constexpr int out_elems = TA * TB * 1024, bias_elems = TA * 32;
  float* dst = rp.partial + ((long)type * nrow + row) * out_elems;
  float* bdst = rp.bias_partial + ((long)type * nrow + row) * bias_elems;
#pragma unroll
  for (int i = 0; i < NI; ++i) {
    const int ta = wa + WA * i;
    if (ta >= TA) continue;
#pragma unroll
    for (int kk = 0; kk < NK; ++kk) {
      const int tb = wb + WB * kk;
      if (tb >= TB) continue;
      f4* d4 = reinterpret_cast<f4*>(dst + ((long)(ta * TB + tb) * 64 + lane) * 16);
#pragma unroll
      for (int q4 = 0; q4 < 4; ++q4) {
        f4 v;
        v[0] = acc[i * NK + kk][4 * q4]; v[1] = acc[i * NK + kk][4 * q4 + 1]; v[2] = acc[i * NK + kk][4 * q4 + 2]; v[3] = acc[i * NK + kk][4 * q4 + 3];
        d4[q4] = v;
      }
    }
    if (wb == i) {
      const float s = xhalf_sum(bsum[i]);
      if (lane < 32) bdst[ta * 32 + lane] = s;
    }
  }
  if (threadIdx.x == 0) { stat_add(rp, 3, t_wait); stat_add(rp, 4, units); stat_add(rp, 5, 1); }
}

// ---------------------------------------------------------------------------------------------------------------- the launch
template <class N>
__global__ __launch_bounds__(64 * BWD_WPB) void mlp_bwd_ring_kernel(BwdArgs a, RingParams rp) {
  extern __shared__ __attribute__((aligned(16))) char lds[];
  __shared__ unsigned s_w[6];   // [0] hand-off slot, [1] abort seen, [2] role ticket / first group, [3] unused, [4], [5] next group (double buffered)
  typedef StageT<BWD_G> ST;
  constexpr AvcOffsets o = Off<N>::value;
  const int xcd = xcc_id();
  if (threadIdx.x == 0) {
    s_w[0] = 0u; s_w[1] = 0u;
    s_w[2] = add_agent(xcd_line(rp, xcd, 0), 1u);
  }
  __syncthreads();
  const int ticket = (int)s_w[2];
  if (ticket < rp.ntypes * rp.cpt) {   // the first arrivals of an XCD are its consumers: product ticket % ntypes, instance ticket / ntypes
    ring_consumer<N>(lds, rp, a, xcd, ticket % rp.ntypes, ticket / rp.ntypes);
    return;
  }
  const int lane0 = threadIdx.x & 63;
  const int wv = threadIdx.x >> 6;
  const long nblk = (a.npts + 31) >> 5;
  ST sg = stage_init<BWD_G>(lds);
  stage_issue(sg, nxt<N, OFF_CHT>(sg, a.Wb0, o), 0);
  const lds_tab_t Tl = tab_to_lds(lds + ST::LDS_BYTES, a.T0, o.v[OFF_TAB_END]);
  const cs_slot_t cs = cs_init<N>(lds + ST::LDS_BYTES + AVC_TAB_LDS_BYTES, wv, lane0);   // column sums of gbar_hs / gbar_h0 (csrc/avc_bwd_body.h)
  __syncthreads();   // (also: every wavefront has read the role ticket)
  if (threadIdx.x == 0) {
    add_agent(xcd_line(rp, xcd, 1), 1u);             // started -- BEFORE the first claim (the consumers' termination test relies on it)
    s_w[4] = add_agent(rp.ctl + RC_NEXT, 1u);
  }
  __syncthreads();
  RingProd<N> ring{rp, xcd, s_w, 0ull, 0ull, 0u};
  unsigned g = s_w[4];
  unsigned long long groups = 0;
  for (int it = 0; g < rp.ngroups; ++it) {
    // the next group is claimed under this one (the returning atomic costs ~1 us); slot (it + 1) & 1 was last read before the
    // barrier that ended iteration it - 1
    if (threadIdx.x == 0) s_w[4 + ((it + 1) & 1)] = add_agent(rp.ctl + RC_NEXT, 1u);
    bwd_sweeps<N>(sg, a, Tl, (long)g * BWD_WPB, nblk, lane0, wv, ring, cs);
    ++groups;
    __syncthreads();
    g = s_w[4 + ((it + 1) & 1)];
    if (s_w[1]) break;
  }
  cs_flush<N>(cs, a.colsum, (long)blockIdx.x * BWD_WPB + wv, lane0);   // (rows of consumer workgroups stay as the caller zeroed them)
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // my arrivals are performed
  __syncthreads();
  if (threadIdx.x == 0) {
    add_agent(xcd_line(rp, xcd, 2), 1u);             // finished
    stat_add(rp, 0, ring.t_alloc); stat_add(rp, 1, ring.t_hand); stat_add(rp, 2, ring.polls); stat_add(rp, 6, groups); stat_add(rp, 7, 1);
  }
}

extern "C" long avc_bwd_ring_ctl_bytes(void) { return (long)RC_WORDS * 4; }
extern "C" long avc_bwd_ring_payload_bytes(int net, int ntypes, int nslots) {
  const int ht = net == AVC_NET_FULL ? NetFull::HT : NetSmall::HT;
  return 8L * ntypes * nslots * BWD_WPB * RING_UNIT_TILES(ht) * 2048;
}
extern "C" int avc_bwd_ring_types(int net) { return net == AVC_NET_FULL ? NetFull::NMID : NetSmall::NMID; }

extern "C" int avc_render_points_bwd_ring(int net, const float* pts, const float* rays_o, const float* rays_d, const float* z,
                                          int S, int ldz, float sample_dist, long npts, const void* wbf16, const float* tab,
                                          const int* offs, const float* d_sdf, const float* d_normal, const float* d_rgb,
                                          const float* rgb_fwd, const void* fpanels, void* gpanels, const void* masks,
                                          float* colsum, void* ctl, void* ring, float* partial, float* bias_partial, const int* pb_tiles,
                                          int ntypes, int cpt, int nslots, int grid, void* stream) {
  if (npts <= 0) return 0;
  if (!fpanels || !gpanels || !masks || !rgb_fwd || !colsum || !ctl || !ring || !partial || !bias_partial || !pb_tiles) {
    avc_set_error("avc_render_points_bwd_ring: NULL buffer");
    return 1;
  }
  if (!(net == AVC_NET_FULL ? offsets_match<NetFull>(offs) : offsets_match<NetSmall>(offs))) {
    avc_set_error("packed-blob offsets differ from the compiled-in table (regenerate csrc/avc_offsets_gen.h)");
    return 1;
  }
  if (ntypes != avc_bwd_ring_types(net) || cpt < 1 || nslots < 1 || nslots > RING_NS_MAX || ntypes > RING_NTY_MAX || grid < 8 * (ntypes * cpt + 1)) {
    avc_set_error("avc_render_points_bwd_ring: ntypes / cpt / nslots / grid out of range");
    return 1;
  }
  if (offs[OFF_TAB_END] * 4 > AVC_TAB_LDS_BYTES) { avc_set_error("fp32 table does not fit its LDS window"); return 1; }
  PointSrc ps{pts, rays_o, rays_d, z, S, ldz, pts ? 0 : 1, sample_dist};
  const long nblk = (npts + 31) / 32;
  RingParams rp;
  rp.ctl = (unsigned*)ctl; rp.ring = (char*)ring; rp.partial = partial; rp.bias_partial = bias_partial;
  rp.ntypes = ntypes; rp.cpt = cpt; rp.nslots = nslots;
  rp.ngroups = (unsigned)((nblk + BWD_WPB - 1) / BWD_WPB);
  for (int t = 0; t < RING_NTY_MAX; ++t) rp.pb[t] = t < ntypes ? pb_tiles[t] : 0;
  hipStream_t s = (hipStream_t)stream;
  if (hipMemsetAsync(ctl, 0, (size_t)RC_WORDS * 4, s) != hipSuccess) { avc_set_error("avc_render_points_bwd_ring: memset failed"); return 1; }
  const int prod_lds = StageT<BWD_G>::LDS_BYTES + AVC_TAB_LDS_BYTES + ColSum<NetFull>::LDS_BYTES, cons_lds = WG_DEPTH * WG_BUF_BYTES;
  const int lds_bytes = prod_lds > cons_lds ? prod_lds : cons_lds;
  static unsigned long long attr_seen = 0;
  if (avc_first_use_on_device(attr_seen)) {
    (void)hipFuncSetAttribute((const void*)mlp_bwd_ring_kernel<NetFull>, hipFuncAttributeMaxDynamicSharedMemorySize, lds_bytes);
    (void)hipFuncSetAttribute((const void*)mlp_bwd_ring_kernel<NetSmall>, hipFuncAttributeMaxDynamicSharedMemorySize, lds_bytes);
  }
  const BwdArgs args{ps, npts, (const b8*)wbf16, tab, d_sdf, d_normal, d_rgb, rgb_fwd, (const char*)fpanels, (char*)gpanels,
                     (const unsigned short*)masks, colsum};
  if (net == AVC_NET_FULL)
    hipLaunchKernelGGL((mlp_bwd_ring_kernel<NetFull>), dim3(grid), dim3(64 * BWD_WPB), lds_bytes, s, args, rp);
  else if (net == AVC_NET_SMALL)
    hipLaunchKernelGGL((mlp_bwd_ring_kernel<NetSmall>), dim3(grid), dim3(64 * BWD_WPB), lds_bytes, s, args, rp);
  else { avc_set_error("unknown net id"); return 1; }
  return avc_check_launch("avc_render_points_bwd_ring");
}
