#include <hip/hip_runtime.h>
#include <cstdio>
#include <cmath>
#include <cstring>
#include <vector>
// stage outputs: 0 u1, 1 s(div), 2 lnf, 3 ln, 4 radius, 5 x, 6 sinx, 7 cosx
__host__ __device__ inline float my_fma(float a, float b, float c) {
#ifdef __HIP_DEVICE_COMPILE__
  return __builtin_fmaf(a,b,c);
#else
  return fmaf(a,b,c);
#endif
}
__host__ __device__ inline void stages(unsigned ra, unsigned rb, float* o) {
  const float u1 = my_fma((float)ra, 0x1p-32f, 0x1p-33f);
  unsigned b; memcpy(&b, &u1, 4);
  int e = (int)(b >> 23) - 127;
  unsigned fb = (b & 0x007FFFFFu) | 0x3F800000u; float f; memcpy(&f, &fb, 4);
  if (f > 1.41421354f) { f = f * 0.5f; e += 1; }
  const float s = (f - 1.0f) / (f + 1.0f);
  const float z = s * s;
  float p = 0.222222224f;
  p = my_fma(p, z, 0.285714298f); p = my_fma(p, z, 0.400000006f); p = my_fma(p, z, 0.666666687f);
  p = p * z;
  const float lnf = my_fma(s, p, s + s);
  const float ef = (float)e;
  const float ln = my_fma(ef, 0.693145751953125f, my_fma(ef, 1.42860677e-06f, lnf));
#ifdef __HIP_DEVICE_COMPILE__
  const float radius = __builtin_sqrtf(-2.0f * ln);
#else
  const float radius = sqrtf(-2.0f * ln);
#endif
  const float t = (float)(rb >> 8) * 0x1p-22f;
  const int q = (int)t; const float fr = t - (float)q;
  const bool swap = fr > 0.5f; const float g = swap ? 1.0f - fr : fr;
  const float x = g * 1.57079637f; const float x2 = x * x;
  float ps = 2.75573188e-06f;
  ps = my_fma(ps, x2, -1.98412701e-04f); ps = my_fma(ps, x2, 8.33333377e-03f); ps = my_fma(ps, x2, -1.66666672e-01f);
  ps = ps * x2;
  const float sinx = my_fma(x, ps, x);
  float pc = -2.75573192e-07f;
  pc = my_fma(pc, x2, 2.48015876e-05f); pc = my_fma(pc, x2, -1.38888892e-03f); pc = my_fma(pc, x2, 4.16666679e-02f); pc = my_fma(pc, x2, -0.5f);
  const float cosx = my_fma(pc, x2, 1.0f);
  o[0]=u1; o[1]=s; o[2]=lnf; o[3]=ln; o[4]=radius; o[5]=x; o[6]=sinx; o[7]=cosx;
}
__global__ void k(const unsigned* ra, const unsigned* rb, float* out, int n) {
  int i = blockIdx.x*blockDim.x+threadIdx.x; if (i<n) stages(ra[i], rb[i], out + 8*(size_t)i);
}
int main(){
  const int n = 1<<20; std::vector<unsigned> ra(n), rb(n);
  unsigned long long st = 88172645463325252ull;
  for (int i=0;i<n;i++){ st ^= st<<13; st ^= st>>7; st ^= st<<17; ra[i]=(unsigned)st; rb[i]=(unsigned)(st>>32);}  
  unsigned *da,*db; float* dout; hipMalloc(&da,n*4); hipMalloc(&db,n*4); hipMalloc(&dout,(size_t)n*32);
  hipMemcpy(da,ra.data(),n*4,hipMemcpyHostToDevice); hipMemcpy(db,rb.data(),n*4,hipMemcpyHostToDevice);
  k<<<n/256,256>>>(da,db,dout,n); std::vector<float> g((size_t)n*8); hipMemcpy(g.data(),dout,(size_t)n*32,hipMemcpyDeviceToHost);
  long bad[8]={0}; int first[8]; for(int j=0;j<8;j++) first[j]=-1;
  for (int i=0;i<n;i++){ float o[8]; stages(ra[i],rb[i],o); for(int j=0;j<8;j++){ if (memcmp(&o[j],&g[(size_t)i*8+j],4)) { bad[j]++; if(first[j]<0) first[j]=i; } } }
  const char* nm[8]={"u1","s","lnf","ln","radius","x","sinx","cosx"};
  for(int j=0;j<8;j++){ printf("%s mismatches %ld", nm[j], bad[j]); if(first[j]>=0){float o[8]; stages(ra[first[j]],rb[first[j]],o); printf(" first i=%d cpu %.9g gpu %.9g", first[j], o[j], g[(size_t)first[j]*8+j]);} printf("\n"); }
  return 0;
}
