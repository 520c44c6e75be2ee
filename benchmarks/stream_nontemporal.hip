// Streaming ceiling with cache-policy hints: the 3-read-1-write triad of stream_ceiling.hip (one tile per workgroup, the
// interpreter's grid shape) with plain, non-temporal-store, non-temporal-load and both variants.
//   hipcc --offload-arch=gfx950 -O3 benchmarks/stream_nontemporal.hip -o /tmp/snt && /tmp/snt
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#define CK(x) do{hipError_t e=(x); if(e!=hipSuccess){printf("err %s line %d\n", hipGetErrorString(e), __LINE__); return 1;}}while(0)
typedef float f32x4 __attribute__((ext_vector_type(4)));
template<bool NTL, bool NTS, int U>
__global__ void __launch_bounds__(256) triad(const f32x4* const* __restrict__ tab, int64_t n4, int tiles_per_row) {
  const f32x4* a = tab[blockIdx.y*4+0]; const f32x4* b = tab[blockIdx.y*4+1]; const f32x4* c = tab[blockIdx.y*4+2]; f32x4* o = (f32x4*)tab[blockIdx.y*4+3];
  for (int tile = blockIdx.x*U; tile < tiles_per_row; tile += gridDim.x*U) {
    f32x4 va[U], vb[U], vc[U];
#pragma unroll
    for (int u=0;u<U;u++){ int64_t i=(int64_t)(tile+u)*256+threadIdx.x; if(i<n4){
      if (NTL) { va[u]=__builtin_nontemporal_load(a+i); vb[u]=__builtin_nontemporal_load(b+i); vc[u]=__builtin_nontemporal_load(c+i); }
      else { va[u]=a[i]; vb[u]=b[i]; vc[u]=c[i]; } } }
#pragma unroll
    for (int u=0;u<U;u++){ int64_t i=(int64_t)(tile+u)*256+threadIdx.x; if(i<n4){ f32x4 r = va[u]+vb[u]+vc[u];
      if (NTS) __builtin_nontemporal_store(r, o+i); else o[i]=r; } }
  }
}
template<bool NTL, bool NTS, int U> float run(const f32x4* const* dtab, int64_t n4, int B, int bpr) {
  int tiles = (int)((n4+255)/256); hipEvent_t e0,e1; hipEventCreate(&e0); hipEventCreate(&e1);
  for(int i=0;i<3;i++) triad<NTL,NTS,U><<<dim3(bpr,B),256>>>(dtab,n4,tiles);
  hipEventRecord(e0); for(int i=0;i<20;i++) triad<NTL,NTS,U><<<dim3(bpr,B),256>>>(dtab,n4,tiles); hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms,e0,e1); return ms/20*1000;
}
int main(){
  const int B=64; const int64_t n=1000000, n4=n/4; std::vector<float*> h(B*4);
  for(int i=0;i<B*4;i++){ CK(hipMalloc(&h[i], n*4+1024)); CK(hipMemset(h[i], 0, n*4)); }
  float** dtab; CK(hipMalloc(&dtab, B*4*8)); CK(hipMemcpy(dtab, h.data(), B*4*8, hipMemcpyHostToDevice));
  const double bytes = 16.0*n*B; const f32x4* const* t = (const f32x4* const*)dtab;
  for (int rep = 0; rep < 2; ++rep)
  for (int bpr : {977, 489, 245}) {
    const int tiles = 977; (void)tiles;
    float p, s, l, ls;
    if (bpr == 977) { p=run<false,false,1>(t,n4,B,bpr); s=run<false,true,1>(t,n4,B,bpr); l=run<true,false,1>(t,n4,B,bpr); ls=run<true,true,1>(t,n4,B,bpr); }
    else if (bpr == 489) { p=run<false,false,2>(t,n4,B,bpr); s=run<false,true,2>(t,n4,B,bpr); l=run<true,false,2>(t,n4,B,bpr); ls=run<true,true,2>(t,n4,B,bpr); }
    else { p=run<false,false,4>(t,n4,B,bpr); s=run<false,true,4>(t,n4,B,bpr); l=run<true,false,4>(t,n4,B,bpr); ls=run<true,true,4>(t,n4,B,bpr); }
    printf("blocks/row %4d: plain %.1f us %.0f GB/s | nt-store %.1f us %.0f | nt-load %.1f us %.0f | both %.1f us %.0f\n", bpr, p, bytes/p/1e3, s, bytes/s/1e3, l, bytes/l/1e3, ls, bytes/ls/1e3);
  }
  return 0;
}
