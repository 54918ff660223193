#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef short s4 __attribute__((ext_vector_type(4)));
typedef short vs4 __attribute__((__vector_size__(4 * sizeof(short))));
// each lane supplies byte address addr[lane]; LDS holds u16 value = its own element index
__global__ void probe(const int* addr, int* out) {
  __shared__ __attribute__((aligned(16))) unsigned short lds[4096];
  for (int i = threadIdx.x; i < 4096; i += 64) lds[i] = (unsigned short)i;
  __syncthreads();
  const int a = addr[threadIdx.x];
  auto p = (__attribute__((address_space(3))) vs4*)((__attribute__((address_space(3))) char*)lds + a);
  vs4 v = __builtin_amdgcn_ds_read_tr16_b64_v4i16(p);
  s4 r = __builtin_bit_cast(s4, v);
  for (int j = 0; j < 4; ++j) out[threadIdx.x * 4 + j] = (unsigned short)r[j];
}
int main() {
  int h_addr[64], *d_addr, *d_out, h_out[256];
  (void)hipMalloc(&d_addr, 256); (void)hipMalloc(&d_out, 1024);
  for (int mode = 0; mode < 2; ++mode) {
    for (int l = 0; l < 64; ++l) h_addr[l] = mode == 0 ? l * 8 : ((l * 37) % 64) * 8 + 1024;
    (void)hipMemcpy(d_addr, h_addr, 256, hipMemcpyHostToDevice);
    probe<<<1, 64>>>(d_addr, d_out);
    (void)hipMemcpy(h_out, d_out, 1024, hipMemcpyDeviceToHost);
    printf("mode %d (lane: supplied element index -> 4 received element indices)\n", mode);
    int ok = 1;
    for (int l = 0; l < 64; ++l) {
      printf("  lane %2d addr_elem %4d -> %4d %4d %4d %4d\n", l, h_addr[l] / 2, h_out[l * 4], h_out[l * 4 + 1], h_out[l * 4 + 2], h_out[l * 4 + 3]);
      for (int j = 0; j < 4; ++j) {
        const int g = l >> 4, i = l & 15;
        const int src_lane = 16 * g + 4 * j + (i >> 2);
        if (h_out[l * 4 + j] != h_addr[src_lane] / 2 + (i & 3)) ok = 0;
      }
    }
    printf("hypothesis (lane i elem j = element (i&3) of the 8 bytes supplied by lane 4j + (i>>2) of its 16-lane group): %s\n", ok ? "HOLDS" : "FAILS");
  }
  return 0;
}
