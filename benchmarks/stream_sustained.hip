// Does the streaming ceiling itself sag under sustained load?  The 3-read-1-write triad of stream_nontemporal.hip (non-temporal
// loads and stores, one tile per workgroup, 64 rows x 1M paths = the bench launch's traffic, no arithmetic to speak of)
// launched back to back for ~2 s, device time per block of 200 launches.  Compared with the same trace of the bench kernel
// (bench.py --steps 2000) it separates "the chip lowers its clocks under this memory load" from "under this VALU load".
//   hipcc --offload-arch=gfx950 -O3 benchmarks/stream_sustained.hip -o /tmp/sus && /tmp/sus
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#define CK(x) do{hipError_t e=(x); if(e!=hipSuccess){printf("err %s line %d\n", hipGetErrorString(e), __LINE__); return 1;}}while(0)
typedef float f32x4 __attribute__((ext_vector_type(4)));
__global__ void __launch_bounds__(256) triad(const f32x4* const* __restrict__ tab, int64_t n4) {
  const f32x4* a = tab[blockIdx.y*4+0]; const f32x4* b = tab[blockIdx.y*4+1]; const f32x4* c = tab[blockIdx.y*4+2]; f32x4* o = (f32x4*)tab[blockIdx.y*4+3];
  const int64_t i = (int64_t)blockIdx.x*256+threadIdx.x;
  if (i < n4) { const f32x4 r = __builtin_nontemporal_load(a+i) + __builtin_nontemporal_load(b+i) + __builtin_nontemporal_load(c+i); __builtin_nontemporal_store(r, o+i); }
}
__global__ void fill_random(float* p, int64_t n, uint32_t seed) {
  for (int64_t i = (int64_t)blockIdx.x*256+threadIdx.x; i < n; i += (int64_t)gridDim.x*256) { uint32_t v = (uint32_t)i*2654435761u ^ seed; v ^= v >> 15; v *= 0x2c1b3c6du; v ^= v >> 12; p[i] = 0.5f + (float)(v >> 8) * (1.0f/16777216.0f); }
}
int main(int argc, char** argv){
  const bool random_data = argc > 1 && argv[1][0] == 'r';        // "r": random operands instead of zeros (data-dependent power / clocks)
  printf("%s operands\n", random_data ? "random" : "zero");
  const int B=64; const int64_t n=1000000, n4=n/4; std::vector<float*> h(B*4);
  for(int i=0;i<B*4;i++){ CK(hipMalloc(&h[i], n*4+1024)); CK(hipMemset(h[i], 0, n*4)); if (random_data) fill_random<<<1024,256>>>(h[i], n, (uint32_t)i*7919u); }
  CK(hipDeviceSynchronize());
  float** dtab; CK(hipMalloc(&dtab, B*4*8)); CK(hipMemcpy(dtab, h.data(), B*4*8, hipMemcpyHostToDevice));
  const double bytes = 16.0*n*B; const f32x4* const* t = (const f32x4* const*)dtab;
  const int blocks = 60, per = 200;
  std::vector<hipEvent_t> ev(blocks+1); for (auto& evt : ev) CK(hipEventCreate(&evt));
  CK(hipEventRecord(ev[0]));
  for (int b = 0; b < blocks; ++b) { for (int i = 0; i < per; ++i) triad<<<dim3(977,B),256>>>(t,n4); CK(hipEventRecord(ev[b+1])); }
  CK(hipDeviceSynchronize());
  double t_acc = 0;
  for (int b = 0; b < blocks; ++b) { float ms; CK(hipEventElapsedTime(&ms, ev[b], ev[b+1])); t_acc += ms; if (b < 8 || b % 6 == 0) printf("t = %6.0f ms: %.1f us per launch, %.0f GB/s\n", t_acc, ms/per*1e3, bytes/(ms/per*1e-3)/1e9); }
  return 0;
}
