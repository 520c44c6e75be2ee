#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#define CK(x) do{hipError_t e=(x); if(e!=hipSuccess){printf("err %s line %d\n", hipGetErrorString(e), __LINE__); return 1;}}while(0)
typedef float f4 __attribute__((ext_vector_type(4)));
// VALU-padded triad: out = f(a,b,c) with NV dependent fma pairs per element to emulate interpreter VALU load.
template<int NV> __device__ __forceinline__ f4 work(f4 a, f4 b, f4 c) {
  f4 r = a + b + c;
#pragma unroll
  for (int i = 0; i < NV; i++) { r = r * c + b; r = r * a + c; }
  return r;
}
// one pass (2 float4 per thread = 2048 elements per block-pass), PASSES per block, with/without prefetch
template<int NV, bool PREFETCH>
__global__ void __launch_bounds__(256) k(const f4* const* __restrict__ tab, unsigned n4, unsigned passes_total, unsigned passes_per_block) {
  const f4* a = tab[blockIdx.y*4+0]; const f4* b = tab[blockIdx.y*4+1]; const f4* c = tab[blockIdx.y*4+2]; f4* o = (f4*)tab[blockIdx.y*4+3];
  unsigned p0 = blockIdx.x * passes_per_block, p1 = min(p0 + passes_per_block, passes_total);
  if (PREFETCH) {
    f4 na[2], nb[2], nc[2];
    { unsigned i = p0*512 + threadIdx.x; for (int t=0;t<2;t++){ unsigned ii=min(i+t*256,n4-1); na[t]=a[ii]; nb[t]=b[ii]; nc[t]=c[ii]; } }
    for (unsigned p = p0; p < p1; ++p) {
      f4 va[2], vb[2], vc[2];
      for (int t=0;t<2;t++){ va[t]=na[t]; vb[t]=nb[t]; vc[t]=nc[t]; }
      if (p + 1 < p1) { unsigned i = (p+1)*512 + threadIdx.x; for (int t=0;t<2;t++){ unsigned ii=min(i+t*256,n4-1); na[t]=a[ii]; nb[t]=b[ii]; nc[t]=c[ii]; } }
      unsigned i = p*512 + threadIdx.x;
      for (int t=0;t<2;t++){ unsigned ii=i+t*256; f4 r = work<NV>(va[t],vb[t],vc[t]); if (ii<n4) o[ii]=r; }
    }
  } else {
    for (unsigned p = p0; p < p1; ++p) {
      unsigned i = p*512 + threadIdx.x; f4 va[2], vb[2], vc[2];
      for (int t=0;t<2;t++){ unsigned ii=min(i+t*256,n4-1); va[t]=a[ii]; vb[t]=b[ii]; vc[t]=c[ii]; }
      for (int t=0;t<2;t++){ unsigned ii=i+t*256; f4 r = work<NV>(va[t],vb[t],vc[t]); if (ii<n4) o[ii]=r; }
    }
  }
}
template<int NV, bool PF> float run(const f4* const* dtab, unsigned n4, int B, unsigned ppb) {
  unsigned passes = (n4 + 511)/512; unsigned bx = (passes + ppb - 1)/ppb; hipEvent_t e0,e1; hipEventCreate(&e0); hipEventCreate(&e1);
  for(int i=0;i<3;i++) k<NV,PF><<<dim3(bx,B),256>>>(dtab,n4,passes,ppb);
  hipEventRecord(e0); for(int i=0;i<10;i++) k<NV,PF><<<dim3(bx,B),256>>>(dtab,n4,passes,ppb); hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms,e0,e1); return ms/10*1000;
}
int main(){
  const int B=64; const int64_t n=1000000; const unsigned n4=n/4; std::vector<float*> h(B*4);
  for(int i=0;i<B*4;i++){ CK(hipMalloc(&h[i], n*4+4096)); CK(hipMemset(h[i], 0, n*4)); }
  float** dtab; CK(hipMalloc(&dtab, B*4*8)); CK(hipMemcpy(dtab, h.data(), B*4*8, hipMemcpyHostToDevice));
  const double bytes = 16.0*n*B;
#define ROW(NV) { printf("NV=%3d (%3d VALU/elem):", NV, 2+4*NV/1); for (unsigned ppb : {1u,2u,4u,8u}) { float a=run<NV,false>((const f4* const*)dtab,n4,B,ppb), b=run<NV,true>((const f4* const*)dtab,n4,B,ppb); printf("  ppb%u: %.0f | pf %.0f us", ppb, a, b);} printf("\n"); }
  ROW(0) ROW(8) ROW(16) ROW(24) ROW(32)
  return 0;
}
