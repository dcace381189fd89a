// Micro-benchmark: HBM read ceilings for the LUT access patterns (tuning aid, not product code).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>
#define CK(x) do{hipError_t e=(x); if(e!=hipSuccess){printf("%s: %s\n",#x,hipGetErrorString(e)); return 1;}}while(0)

template<int KSUB> __global__ __launch_bounds__(256) void k_dword(const int* __restrict__ in, size_t n, unsigned* out){
  const unsigned lane = threadIdx.x & 63u;
  const unsigned wave0 = (blockIdx.x*256+threadIdx.x)>>6, nw = gridDim.x*4;
  const unsigned nch = n/(64*KSUB);
  unsigned acc=0;
  for(unsigned ch=wave0; ch<nch; ch+=nw){
    const int* p = in + (size_t)ch*64*KSUB + lane;
    int v[KSUB];
    #pragma unroll
    for(int k=0;k<KSUB;++k) v[k]=p[64*k];
    #pragma unroll
    for(int k=0;k<KSUB;++k) acc += v[k];
  }
  if(acc==0x12345678) out[0]=acc;
}
template<int KV> __global__ __launch_bounds__(256) void k_x4(const int4* __restrict__ in, size_t n4, unsigned* out){
  const unsigned lane = threadIdx.x & 63u;
  const unsigned wave0 = (blockIdx.x*256+threadIdx.x)>>6, nw = gridDim.x*4;
  const unsigned nch = n4/(64*KV);
  unsigned acc=0;
  for(unsigned ch=wave0; ch<nch; ch+=nw){
    const int4* p = in + (size_t)ch*64*KV + lane;
    int4 v[KV];
    #pragma unroll
    for(int k=0;k<KV;++k) v[k]=p[64*k];
    #pragma unroll
    for(int k=0;k<KV;++k) acc += v[k].x+v[k].y+v[k].z+v[k].w;
  }
  if(acc==0x12345678) out[0]=acc;
}
// simple: one thread per element(s), huge grid
__global__ __launch_bounds__(256) void k_flat_x4(const int4* __restrict__ in, size_t n4, unsigned* out){
  size_t i=(size_t)blockIdx.x*256+threadIdx.x; unsigned acc=0;
  if(i<n4){int4 v=in[i]; acc=v.x+v.y+v.z+v.w;}
  if(acc==0x12345678) out[0]=acc;
}
int main(){
  const size_t n = (size_t)1<<32;   // 4G ints = 17.2 GB
  int* d; unsigned* o; CK(hipMalloc(&d, n*4)); CK(hipMalloc(&o,4));
  CK(hipMemset(d, 1, n*4));
  hipEvent_t a,b; hipEventCreate(&a); hipEventCreate(&b);
  auto run=[&](const char* name, auto launch){
    launch(); hipDeviceSynchronize(); float best=1e9;
    for(int r=0;r<5;++r){ hipEventRecord(a); launch(); hipEventRecord(b); hipEventSynchronize(b); float ms; hipEventElapsedTime(&ms,a,b); if(ms<best)best=ms; }
    printf("%-28s %.3f ms  %.2f TB/s\n", name, best, n*4/best/1e9);
  };
  for (int blocks : {2048, 4096, 8192}) {
    printf("grid %d blocks\n", blocks);
    run("dword KSUB=4", [&]{ k_dword<4><<<blocks,256>>>(d,n,o); });
    run("dword KSUB=8", [&]{ k_dword<8><<<blocks,256>>>(d,n,o); });
    run("dword KSUB=16", [&]{ k_dword<16><<<blocks,256>>>(d,n,o); });
    run("dwordx4 KV=1", [&]{ k_x4<1><<<blocks,256>>>((int4*)d,n/4,o); });
    run("dwordx4 KV=2", [&]{ k_x4<2><<<blocks,256>>>((int4*)d,n/4,o); });
    run("dwordx4 KV=4", [&]{ k_x4<4><<<blocks,256>>>((int4*)d,n/4,o); });
  }
  run("flat dwordx4 1/thread", [&]{ k_flat_x4<<<(unsigned)(n/4/256),256>>>((int4*)d,n/4,o); });
  return 0;
}
