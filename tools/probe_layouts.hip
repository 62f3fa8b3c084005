// Hardware-layout probe for gfx950: pins down the MFMA operand/result lane maps and the
// ds_read_b64_tr_b16 gather that the kernels in uvc_amd/csrc rely on.  Test tool only.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cmath>
#include <vector>
typedef short s4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
#define LDS3(T) __attribute__((address_space(3))) T

__device__ inline unsigned short f2bf(float f){ unsigned u=__float_as_uint(f); u += 0x7fff + ((u>>16)&1); return u>>16; }

__global__ void k_tr_linear(short* out){
  __shared__ short lds[2048];
  for(int i=threadIdx.x;i<2048;i+=64) lds[i]=i;
  __syncthreads();
  s4 v = __builtin_amdgcn_ds_read_tr16_b64_v4i16((LDS3(s4)*)(lds + threadIdx.x*4));
  for(int j=0;j<4;j++) out[threadIdx.x*4+j]=v[j];
}
// lane i of a 16-lane group supplies &M[(i>>2)*ld + (i&3)*4]; group g block at row 4*g
__global__ void k_tr_strided(short* out, int ld){
  __shared__ short lds[4096];
  for(int i=threadIdx.x;i<4096;i+=64) lds[i]=i;
  __syncthreads();
  int l=threadIdx.x, i=l&15, g=l>>4;
  s4 v = __builtin_amdgcn_ds_read_tr16_b64_v4i16((LDS3(s4)*)(lds + (g*4+(i>>2))*ld + (i&3)*4));
  for(int j=0;j<4;j++) out[l*4+j]=v[j];
}
// bf16 16x16x32: A[16][32] row-major, B[32][16] row-major -> D[16][16]
__global__ void k_mfma_bf16(const float* A, const float* B, float* D){
  int l=threadIdx.x;
  bf16x8 a,b;
  for(int s=0;s<8;s++){
    int k=(l>>4)*8+s;
    unsigned short ua=f2bf(A[(l&15)*32+k]), ub=f2bf(B[k*16+(l&15)]);
    a[s]=__builtin_bit_cast(__bf16, ua); b[s]=__builtin_bit_cast(__bf16, ub);
  }
  f32x4 c={0,0,0,0};
  c=__builtin_amdgcn_mfma_f32_16x16x32_bf16(a,b,c,0,0,0);
  for(int r=0;r<4;r++) D[((l>>4)*4+r)*16+(l&15)]=c[r];
}
__global__ void k_mfma_f32(const float* A, const float* B, float* D){ // A[16][4], B[4][16]
  int l=threadIdx.x;
  float a=A[(l&15)*4+(l>>4)], b=B[(l>>4)*16+(l&15)];
  f32x4 c={0,0,0,0};
  c=__builtin_amdgcn_mfma_f32_16x16x4f32(a,b,c,0,0,0);
  for(int r=0;r<4;r++) D[((l>>4)*4+r)*16+(l&15)]=c[r];
}
// is fp32 division correctly rounded, and does fp64 work as expected
__global__ void k_div(const float* a, const float* b, float* q, int n){
  int i=threadIdx.x+blockIdx.x*blockDim.x; if(i<n) q[i]=a[i]/b[i];
}
int main(){
  int fails=0;
  { short* d; hipMalloc(&d,64*4*2); k_tr_linear<<<1,64>>>(d); std::vector<short> h(256); hipMemcpy(h.data(),d,512,hipMemcpyDeviceToHost);
    int bad=0; for(int l=0;l<64;l++)for(int j=0;j<4;j++){int e=(l&15)+j*16+(l>>4)*64; if(h[l*4+j]!=e) bad++;}
    printf("tr_linear hypothesis lds[(l&15)+j*16+(l>>4)*64]: %s (bad=%d)\n", bad?"FAIL":"PASS",bad); fails+=bad!=0;
    if(bad){ for(int l=0;l<64;l++){printf("lane %2d:",l); for(int j=0;j<4;j++)printf(" %4d",h[l*4+j]); printf("\n");} }
  }
  for(int ld: {16,24,72,136}){ short* d; hipMalloc(&d,512); k_tr_strided<<<1,64>>>(d,ld); std::vector<short> h(256); hipMemcpy(h.data(),d,512,hipMemcpyDeviceToHost);
    int bad=0; for(int l=0;l<64;l++)for(int j=0;j<4;j++){int i=l&15,g=l>>4; int e=(g*4+j)*ld+i; if(h[l*4+j]!=e) bad++;}
    printf("tr_strided ld=%d hypothesis out[l][j]=M[4g+j][i]: %s (bad=%d)\n", ld, bad?"FAIL":"PASS",bad); fails+=bad!=0;
    if(bad){ for(int l=0;l<64;l++){printf("lane %2d:",l); for(int j=0;j<4;j++)printf(" %4d",h[l*4+j]); printf("\n");} }
  }
  { std::vector<float> A(512),B(512),D(256),R(256,0.f); srand(1);
    for(auto&x:A)x=(rand()%17-8); for(auto&x:B)x=(rand()%13-6);
    for(int i=0;i<16;i++)for(int j=0;j<16;j++){float s=0;for(int k=0;k<32;k++)s+=A[i*32+k]*B[k*16+j];R[i*16+j]=s;}
    float *dA,*dB,*dD; hipMalloc(&dA,2048);hipMalloc(&dB,2048);hipMalloc(&dD,1024);
    hipMemcpy(dA,A.data(),2048,hipMemcpyHostToDevice);hipMemcpy(dB,B.data(),2048,hipMemcpyHostToDevice);
    k_mfma_bf16<<<1,64>>>(dA,dB,dD); hipMemcpy(D.data(),dD,1024,hipMemcpyDeviceToHost);
    int bad=0; for(int i=0;i<256;i++) if(D[i]!=R[i]) bad++;
    printf("mfma_bf16_16x16x32 layout: %s (bad=%d)\n", bad?"FAIL":"PASS",bad); fails+=bad!=0; }
  { std::vector<float> A(64),B(64),D(256),R(256,0.f); srand(2);
    for(auto&x:A)x=(rand()%17-8)*0.37f; for(auto&x:B)x=(rand()%13-6)*1.13f;
    for(int i=0;i<16;i++)for(int j=0;j<16;j++){float s=0;for(int k=0;k<4;k++)s=fmaf(A[i*4+k],B[k*16+j],s);R[i*16+j]=s;}
    float *dA,*dB,*dD; hipMalloc(&dA,256);hipMalloc(&dB,256);hipMalloc(&dD,1024);
    hipMemcpy(dA,A.data(),256,hipMemcpyHostToDevice);hipMemcpy(dB,B.data(),256,hipMemcpyHostToDevice);
    k_mfma_f32<<<1,64>>>(dA,dB,dD); hipMemcpy(D.data(),dD,1024,hipMemcpyDeviceToHost);
    int bad=0; for(int i=0;i<256;i++) if(D[i]!=R[i]) bad++;
    printf("mfma_f32_16x16x4 layout + fmaf-chain bitwise: %s (bad=%d)\n", bad?"FAIL":"PASS",bad); fails+=bad!=0; }
  { int n=4096; std::vector<float> a(n),b(n),q(n); srand(3); for(int i=0;i<n;i++){a[i]=(rand()/(float)RAND_MAX-0.5f)*0.1f; b[i]=1.0f+rand()/(float)RAND_MAX*1e-3f;}
    float *da,*db,*dq; hipMalloc(&da,n*4);hipMalloc(&db,n*4);hipMalloc(&dq,n*4);
    hipMemcpy(da,a.data(),n*4,hipMemcpyHostToDevice);hipMemcpy(db,b.data(),n*4,hipMemcpyHostToDevice);
    k_div<<<n/256,256>>>(da,db,dq,n); hipMemcpy(q.data(),dq,n*4,hipMemcpyDeviceToHost);
    int bad=0; for(int i=0;i<n;i++){ volatile float r=a[i]/b[i]; if(q[i]!=r) bad++; }
    printf("fp32 divide correctly rounded vs host: %s (bad=%d)\n", bad?"FAIL":"PASS",bad); fails+=bad!=0; }
  hipDeviceProp_t p; hipGetDeviceProperties(&p,0);
  printf("device %s CUs=%d clock=%d MHz smem/block=%zu\n", p.gcnArchName, p.multiProcessorCount, p.clockRate/1000, p.sharedMemPerBlock);
  printf("PROBE %s\n", fails?"HAS_FAILURES":"ALL_PASS");
  return 0;
}
