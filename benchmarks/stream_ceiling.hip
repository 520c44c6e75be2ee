#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#define CK(x) do{hipError_t e=(x); if(e!=hipSuccess){printf("err %s line %d\n", hipGetErrorString(e), __LINE__); return 1;}}while(0)
// out = a+b+c over B rows via pointer table; TILES float4 per thread per iteration
template<int U, int WPS>
__global__ void __launch_bounds__(256, WPS) triad(const float4* const* __restrict__ tab, int64_t n4, int tiles_per_row) {
  const float4* a = tab[blockIdx.y*4+0]; const float4* b = tab[blockIdx.y*4+1]; const float4* c = tab[blockIdx.y*4+2]; float4* o = (float4*)tab[blockIdx.y*4+3];
  for (int tile = blockIdx.x*U; tile < tiles_per_row; tile += gridDim.x*U) {
    float4 va[U], vb[U], vc[U];
#pragma unroll
    for (int u=0;u<U;u++){ int64_t i=(int64_t)(tile+u)*256+threadIdx.x; if(i<n4){ va[u]=a[i]; vb[u]=b[i]; vc[u]=c[i]; } }
#pragma unroll
    for (int u=0;u<U;u++){ int64_t i=(int64_t)(tile+u)*256+threadIdx.x; if(i<n4){ float4 r; r.x=va[u].x+vb[u].x+vc[u].x; r.y=va[u].y+vb[u].y+vc[u].y; r.z=va[u].z+vb[u].z+vc[u].z; r.w=va[u].w+vb[u].w+vc[u].w; o[i]=r; } }
  }
}
template<int U, int WPS> float run(const float4* const* dtab, int64_t n4, int B, int bpr) {
  int tiles = (int)((n4+255)/256); hipEvent_t e0,e1; hipEventCreate(&e0); hipEventCreate(&e1);
  for(int i=0;i<3;i++) triad<U,WPS><<<dim3(bpr,B),256>>>(dtab,n4,tiles);
  hipEventRecord(e0); for(int i=0;i<10;i++) triad<U,WPS><<<dim3(bpr,B),256>>>(dtab,n4,tiles); hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms,e0,e1); return ms/10*1000;
}
int main(){
  const int B=64; const int64_t n=1000000, n4=n/4; std::vector<float*> h(B*4);
  for(int i=0;i<B*4;i++){ CK(hipMalloc(&h[i], n*4+1024)); CK(hipMemset(h[i], 0, n*4)); }
  float** dtab; CK(hipMalloc(&dtab, B*4*8)); CK(hipMemcpy(dtab, h.data(), B*4*8, hipMemcpyHostToDevice));
  const double bytes = 16.0*n*B;
  for (int bpr : {16, 32, 64, 128, 256, 977}) {
    float a=run<1,4>((const float4* const*)dtab,n4,B,bpr), b=run<1,8>((const float4* const*)dtab,n4,B,bpr), c=run<2,4>((const float4* const*)dtab,n4,B,bpr), d=run<2,8>((const float4* const*)dtab,n4,B,bpr), e=run<4,4>((const float4* const*)dtab,n4,B,bpr);
    printf("bpr %4d: U1/w4 %.1f us %.0f GB/s | U1/w8 %.1f us %.0f | U2/w4 %.1f us %.0f | U2/w8 %.1f %.0f | U4/w4 %.1f %.0f\n", bpr, a, bytes/a/1e3, b, bytes/b/1e3, c, bytes/c/1e3, d, bytes/d/1e3, e, bytes/e/1e3);
  }
  return 0;
}
