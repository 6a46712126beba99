// Developer aid: where does the mul+FMA divide differ from the IEEE divide on gfx950?
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <cstring>
__device__ __forceinline__ float div5(float a, float d, float rd){ float q=a*rd; float e=fmaf(-q,d,a); q=fmaf(e,rd,q); e=fmaf(-q,d,a); return fmaf(e,rd,q);}
__device__ __forceinline__ float div3(float a, float d, float rd){ float q=a*rd; float e=fmaf(-q,d,a); return fmaf(e,rd,q);}
__global__ void chk(float d, float rd, uint32_t e_lo, uint32_t e_hi, unsigned long long* cnt, uint32_t* ex){
  const uint64_t total=(uint64_t)(e_hi-e_lo+1)<<23;
  for(uint64_t t=(uint64_t)blockIdx.x*blockDim.x+threadIdx.x;t<total;t+=(uint64_t)gridDim.x*blockDim.x){
    uint32_t bits=((uint32_t)(e_lo+(uint32_t)(t>>23))<<23)|(uint32_t)(t&((1u<<23)-1));
    float a=__uint_as_float(bits);
    float r=a/d; float f5=div5(a,d,rd), f3=div3(a,d,rd);
    if(__float_as_uint(r)!=__float_as_uint(f5)){ unsigned long long k=atomicAdd(&cnt[0],1ull); if(k<8){ex[3*k]=bits;ex[3*k+1]=__float_as_uint(r);ex[3*k+2]=__float_as_uint(f5);} }
    if(__float_as_uint(r)!=__float_as_uint(f3)) atomicAdd(&cnt[1],1ull);
  }
}
int main(){
  unsigned long long* cnt; uint32_t* ex; hipMalloc(&cnt,16); hipMalloc(&ex,96);
  float ds[2]={4000.0f,0.05f}; uint32_t lo[2]={127,27}, hi[2]={159,167};
  for(int k=0;k<2;k++){ hipMemset(cnt,0,16); hipMemset(ex,0,96);
    chk<<<4096,256>>>(ds[k],1.0f/ds[k],lo[k],hi[k],cnt,ex); hipDeviceSynchronize();
    unsigned long long h[2]; uint32_t hx[24]; hipMemcpy(h,cnt,16,hipMemcpyDeviceToHost); hipMemcpy(hx,ex,96,hipMemcpyDeviceToHost);
    printf("d=%g bad5=%llu bad3=%llu\n",ds[k],h[0],h[1]);
    for(int i=0;i<8&&i<(int)h[0];i++){ float a,r,f; memcpy(&a,&hx[3*i],4); memcpy(&r,&hx[3*i+1],4); memcpy(&f,&hx[3*i+2],4); printf("  a=%08x (%g) ieee=%08x fast=%08x\n",hx[3*i],a,hx[3*i+1],hx[3*i+2]); }
  }
  return 0;
}
