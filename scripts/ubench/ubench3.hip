// Micro-benchmark, round 3 (development aid; results in profiles/r03_ubench3_*.txt) -- VERDICT r2 item 1(i):
// can gradient-type operand tiles travel from a producing workgroup to a consuming workgroup through an L2-resident ring instead
// of through HBM?  Producer workgroups (8 wavefronts, one 2-KiB tile per wavefront and step = the store pattern of mlp_bwd_kernel)
// write slots of a ring, consumer workgroups of the SAME XCD (paired by the hardware XCC id, so the protocol does not assume a
// block -> XCD map) read them behind a flag.
//   mode 0  same-XCD hand-off without any L2 write-back: plain stores -> s_waitcnt vmcnt(0) -> workgroup barrier -> lane 0 publishes
//           a relaxed agent-scope counter; the consumer polls it (sc1 load), then reads the tiles with sc1 loads (L1 bypassed, served
//           by the XCD's L2, where the producer's lines are).  Legal ONLY because both ends share an L2.
//   mode 1  the placement-independent protocol of the guide (producer: agent-scope release = buffer_wbl2; consumer: agent-scope
//           acquire, plain loads): what a cross-XCD pipeline would have to pay.
// Every word is checked (value = f(pair, step, tile, lane)); every spin is bounded (a mis-pairing ends the kernel, it cannot hang).
// Run under rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE to see whether the ring bytes reach HBM.
#include <hip/hip_runtime.h>
#include <stdint.h>
typedef int v4i __attribute__((ext_vector_type(4)));

__device__ __forceinline__ int xcc_id() { return __builtin_amdgcn_s_getreg((3 << 11) | (0 << 6) | 20) & 0xf; }

struct Ring {
  v4i* data;            // [xcd][pair][slot][8 tiles][128 x 16 B]
  unsigned* prod;       // [xcd][pair] steps published   (one 128-B line each)
  unsigned* cons;       // [xcd][pair] steps consumed
  unsigned* ticket;     // [xcd] role tickets
  unsigned* errors;     // [0] payload mismatches, [1] spin time-outs, [2] unpaired workgroups
  int pairs_per_xcd, slots, steps;
};

__device__ __forceinline__ unsigned poll(const unsigned* p) {
  return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);   // global_load sc1: bypasses L1
}
__device__ __forceinline__ v4i pattern(unsigned pair, unsigned step, unsigned tile, unsigned q) {
  v4i v;
  v.x = (int)(pair * 2654435761u + step);
  v.y = (int)(step * 40503u + tile);
  v.z = (int)q;
  v.w = (int)(pair ^ (step << 8) ^ (tile << 20));
  return v;
}

// SLOTS is a template parameter only to give every ring size its own kernel name in the rocprofv3 tables
template <int MODE, int SLOTS>
__global__ __launch_bounds__(512) void ub3_ring_kernel(Ring r) {
  const int xcd = xcc_id();
  __shared__ unsigned s_ticket;
  if (threadIdx.x == 0) s_ticket = atomicAdd(&r.ticket[xcd], 1u);
  __syncthreads();
  const unsigned ticket = s_ticket;
  const unsigned pair = ticket >> 1;
  const bool producer = (ticket & 1u) == 0;
  if ((int)pair >= r.pairs_per_xcd) { if (threadIdx.x == 0) atomicAdd(&r.errors[2], 1u); return; }
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const long pidx = (long)xcd * r.pairs_per_xcd + pair;
  v4i* ring = r.data + pidx * r.slots * (8 * 128);
  unsigned* pflag = r.prod + pidx * 32;
  unsigned* cflag = r.cons + pidx * 32;
  __shared__ int s_abort;
  if (threadIdx.x == 0) s_abort = 0;
  __syncthreads();
  unsigned bad = 0;
  for (int step = 0; step < r.steps; ++step) {
    v4i* tile = ring + (long)(step % r.slots) * (8 * 128) + wave * 128;
    if (producer) {
      // the slot is free once the consumer is done with step - slots
      if (threadIdx.x == 0 && step >= r.slots) {
        int spins = 0;
        while (poll(cflag) < (unsigned)(step - r.slots + 1)) {
          __builtin_amdgcn_s_sleep(2);
          if (++spins > 4000000) { atomicAdd(&r.errors[1], 1u); s_abort = 1; break; }
        }
      }
      __syncthreads();
      if (s_abort) return;
      tile[lane] = pattern((unsigned)pidx, step, wave, lane);            // 2 x 1 KiB per wavefront: a panel tile
      tile[64 + lane] = pattern((unsigned)pidx, step, wave, 64 + lane);
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");                    // every wavefront's stores have reached the XCD's L2
      __syncthreads();
      if (threadIdx.x == 0) {
        if (MODE == 1) {
          __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");              // buffer_wbl2 sc1: the dirty lines leave the L2
          asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        }
        __hip_atomic_store(pflag, (unsigned)(step + 1), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      }
    } else {
      if (threadIdx.x == 0) {
        int spins = 0;
        while (poll(pflag) < (unsigned)(step + 1)) {
          __builtin_amdgcn_s_sleep(2);
          if (++spins > 4000000) { atomicAdd(&r.errors[1], 1u); s_abort = 1; break; }
        }
        if (MODE == 1) __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
      }
      __syncthreads();
      if (s_abort) return;
      v4i a, b;
      if (MODE == 0) {   // L1-bypassing loads (sc1), served by the L2 the producer wrote into
        const v4i* t0 = tile + lane;
        const v4i* t1 = tile + 64 + lane;
        asm volatile("global_load_dwordx4 %0, %1, off sc1" : "=v"(a) : "v"(t0) : "memory");
        asm volatile("global_load_dwordx4 %0, %1, off sc1" : "=v"(b) : "v"(t1) : "memory");
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      } else {
        a = tile[lane];
        b = tile[64 + lane];
      }
      const v4i ea = pattern((unsigned)pidx, step, wave, lane), eb = pattern((unsigned)pidx, step, wave, 64 + lane);
      bad += (a.x != ea.x) | (a.y != ea.y) | (a.z != ea.z) | (a.w != ea.w);
      bad += (b.x != eb.x) | (b.y != eb.y) | (b.z != eb.z) | (b.w != eb.w);
      __syncthreads();
      if (threadIdx.x == 0) __hip_atomic_store(cflag, (unsigned)(step + 1), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
  }
  if (bad) atomicAdd(&r.errors[0], bad);
}

extern "C" int ub3_ring(int mode, int grid, void* data, void* prod, void* cons, void* ticket, void* errors, int pairs_per_xcd,
                        int slots, int steps, void* stream) {
  Ring r{(v4i*)data, (unsigned*)prod, (unsigned*)cons, (unsigned*)ticket, (unsigned*)errors, pairs_per_xcd, slots, steps};
#define UB3_L(M, S) hipLaunchKernelGGL((ub3_ring_kernel<M, S>), dim3(grid), dim3(512), 0, (hipStream_t)stream, r)
  if (mode == 0 && slots == 4) UB3_L(0, 4);
  else if (mode == 0 && slots == 8) UB3_L(0, 8);
  else if (mode == 0 && slots == 64) UB3_L(0, 64);
  else if (mode == 1 && slots == 8) UB3_L(1, 8);
  else return -1;
  return (int)hipGetLastError();
}

// reference point: the same tiles written to / read from a buffer far larger than every cache, no hand-off (what the panels do today)
__global__ __launch_bounds__(512) void ub3_stream_kernel(v4i* buf, long tiles_per_wg, int write) {
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  v4i* base = buf + (long)blockIdx.x * tiles_per_wg * 128;
  unsigned acc = 0;
  for (long t = wave; t < tiles_per_wg; t += 8) {
    v4i* tile = base + t * 128;
    if (write == 2) {          // normal-policy stores (stay in the XCD's write-back L2 until evicted)
      tile[lane] = pattern(blockIdx.x, (unsigned)t, wave, lane);
      tile[64 + lane] = pattern(blockIdx.x, (unsigned)t, wave, 64 + lane);
    } else if (write) {
      __builtin_nontemporal_store(pattern(blockIdx.x, (unsigned)t, wave, lane), tile + lane);
      __builtin_nontemporal_store(pattern(blockIdx.x, (unsigned)t, wave, 64 + lane), tile + 64 + lane);
    } else {
      v4i a = __builtin_nontemporal_load(tile + lane), b = __builtin_nontemporal_load(tile + 64 + lane);
      acc += a.x ^ b.y;
    }
  }
  if (acc == 0x12345678u) buf[0].x = 1;
}
extern "C" int ub3_stream(int grid, void* buf, long tiles_per_wg, int write, void* stream) {
  hipLaunchKernelGGL(ub3_stream_kernel, dim3(grid), dim3(512), 0, (hipStream_t)stream, (v4i*)buf, tiles_per_wg, write);
  return (int)hipGetLastError();
}
