// TEST INFRASTRUCTURE ONLY -- lockstep wavefront emulator.
//
// Lets the *unmodified* product sources (csrc/wrsn_api.hip + csrc/wrsn_sim.h) be compiled by g++ and
// executed on the CPU so the kernel logic can be checked against the oracle in `-m "not gpu"` tests.
// It is not a backend: the product library (libwrsn_hip.so, built by hipcc for gfx950) never sees this
// header, and nothing in the product package can load the emulated library.
//
// Model: the threads of one workgroup are fibers on one OS thread, run round-robin; every
// collective (__syncthreads, __shfl*, __ballot) is a rendezvous of all fibers of the block, so
// wave-uniform code behaves as on the 64-lane hardware wavefront.  Blocks run one after another.
#pragma once
#include <cmath>
#include <cstddef>
#include <cstdint>
#include <cstdlib>
#include <cstring>
#include <functional>

#define __global__
#define WRSN_SORT_THREADS 256                  /* at most 256 fibers per block here */
#define __device__
#define __host__
#define __shared__
#define __forceinline__ inline __attribute__((always_inline))
#define __launch_bounds__(...)
#define __noinline__ __attribute__((noinline))
inline void __threadfence() {}
#define WRSN_GLOBAL_AS                       /* one address space here */
#define WRSN_LD_U4_DEFINED
template <typename T> inline T wrsn_ld_u4(const T* p) { return *p; }

struct alignas(16) float4 { float x, y, z, w; };
struct dim3 { unsigned x, y, z; dim3(unsigned x_ = 1, unsigned y_ = 1, unsigned z_ = 1) : x(x_), y(y_), z(z_) {} };
struct emu_idx { unsigned x, y, z; };
extern emu_idx threadIdx, blockIdx, blockDim, gridDim;
extern double smem[];                        // dynamic LDS of the running block

typedef int hipError_t;
typedef void* hipStream_t;
enum { hipSuccess = 0, hipErrorUnknown = 1 };
enum hipMemcpyKind { hipMemcpyHostToDevice, hipMemcpyDeviceToHost, hipMemcpyDeviceToDevice };
inline const char* hipGetErrorString(hipError_t) { return "emulated"; }
inline hipError_t hipGetDeviceCount(int* n) { *n = 1; return hipSuccess; }
inline hipError_t hipSetDevice(int) { return hipSuccess; }
inline hipError_t hipGetDevice(int* d) { *d = 0; return hipSuccess; }
inline hipError_t hipMalloc(void** p, size_t n) { *p = std::malloc(n); return *p ? hipSuccess : hipErrorUnknown; }
inline hipError_t hipFree(void* p) { std::free(p); return hipSuccess; }
inline hipError_t hipMemset(void* p, int v, size_t n) { std::memset(p, v, n); return hipSuccess; }
inline unsigned long long atomicCAS(unsigned long long* p, unsigned long long cmp, unsigned long long v) { const unsigned long long o = *p; if (o == cmp) *p = v; return o; }
struct hipDeviceProp_t { int multiProcessorCount; };
inline hipError_t hipGetDeviceProperties(hipDeviceProp_t* p, int) { p->multiProcessorCount = 256; return hipSuccess; }
template <typename F> inline hipError_t hipOccupancyMaxActiveBlocksPerMultiprocessor(int* n, F, int, size_t) { *n = 8; return hipSuccess; }
inline hipError_t hipMemsetAsync(void* p, int v, size_t n, hipStream_t) { std::memset(p, v, n); return hipSuccess; }
inline hipError_t hipMemcpy(void* d, const void* s, size_t n, hipMemcpyKind) { std::memmove(d, s, n); return hipSuccess; }
inline hipError_t hipMemcpyAsync(void* d, const void* s, size_t n, hipMemcpyKind, hipStream_t) { std::memmove(d, s, n); return hipSuccess; }
inline hipError_t hipStreamSynchronize(hipStream_t) { return hipSuccess; }
typedef void* hipEvent_t;
inline hipError_t hipEventCreate(hipEvent_t* e) { *e = nullptr; return hipSuccess; }
inline hipError_t hipEventDestroy(hipEvent_t) { return hipSuccess; }
inline hipError_t hipEventRecord(hipEvent_t, hipStream_t) { return hipSuccess; }
inline hipError_t hipEventElapsedTime(float* ms, hipEvent_t, hipEvent_t) { *ms = 0.f; return hipSuccess; }
inline hipError_t hipDeviceSynchronize() { return hipSuccess; }
enum { hipStreamNonBlocking = 1, hipEventDisableTiming = 2 };
inline hipError_t hipStreamCreateWithFlags(hipStream_t* s, unsigned) { *s = nullptr; return hipSuccess; }   /* kernels run at their launch call, one after another */
inline hipError_t hipStreamDestroy(hipStream_t) { return hipSuccess; }
inline hipError_t hipDeviceGetStreamPriorityRange(int* lo, int* hi) { *lo = 0; *hi = 0; return hipSuccess; }
inline hipError_t hipStreamCreateWithPriority(hipStream_t* s, unsigned, int) { *s = nullptr; return hipSuccess; }
inline hipError_t hipEventCreateWithFlags(hipEvent_t* e, unsigned) { *e = nullptr; return hipSuccess; }
inline hipError_t hipStreamWaitEvent(hipStream_t, hipEvent_t, unsigned) { return hipSuccess; }
inline hipError_t hipGetLastError() { return hipSuccess; }

// ---- rendezvous primitives implemented in emu_runtime.cpp
void emu_barrier();
void emu_wave_barrier();
extern unsigned char emu_slots[2][256][16];
extern int emu_parity;
void emu_run_block(const std::function<void()>& body, unsigned nthreads);

inline void __syncthreads() { emu_barrier(); }

template <typename T>
inline T emu_exchange(T v, int src_lane) {
    static_assert(sizeof(T) <= 16, "shuffle payload");
    const int p = emu_parity;                // same value in every fiber between two rendezvous
    std::memcpy(emu_slots[p][threadIdx.x], &v, sizeof(T));
    emu_barrier();                           // flips emu_parity once all lanes have arrived
    T r;
    std::memcpy(&r, emu_slots[p][(threadIdx.x & ~63u) | (unsigned)(src_lane & 63)], sizeof(T));   // the source lane of the caller's own wavefront
    return r;
}
/* no wall clock here: lane 0 of the wavefront advances a counter and every lane takes ITS reading (a rendezvous): wave-uniform like s_memrealtime */
inline long long wall_clock64() { static long long t_ = 0; long long v = 0; if ((threadIdx.x & 63u) == 0) v = (t_ += 50); return emu_exchange(v, 0); }
template <typename T> inline T __shfl(T v, int src) { return emu_exchange(v, src); }
template <typename T> inline T __shfl_xor(T v, int mask) { return emu_exchange(v, (int)threadIdx.x ^ mask); }
template <typename T> inline T __shfl_up(T v, int delta) { int s = (int)threadIdx.x - delta; return emu_exchange(v, s < 0 ? (int)threadIdx.x : s); }
inline unsigned long long __ballot(int pred) {
    const int p = emu_parity;
    unsigned char b = pred ? 1 : 0;
    emu_slots[p][threadIdx.x][0] = b;
    emu_barrier();
    unsigned long long m = 0;
    for (unsigned l = 0; l < blockDim.x && l < 64; ++l) if (emu_slots[p][l][0]) m |= 1ull << l;
    return m;
}

// ---- DPP / readlane builtins used by the wave reductions (only the controls the product uses)
inline int emu_update_dpp(int old, int src, int ctrl, int row_mask, int /*bank_mask*/, bool /*bound_ctrl*/) {
    const int l = (int)threadIdx.x, row = l >> 4;
    int from = -1;
    switch (ctrl) {
    case 0xB1: from = (l & ~3) | ((l & 3) ^ 1); break;
    case 0x4E: from = (l & ~3) | ((l & 3) ^ 2); break;
    case 0x141: from = (l & ~7) | (7 - (l & 7)); break;
    case 0x140: from = (l & ~15) | (15 - (l & 15)); break;
    case 0x142: from = row >= 1 ? (row - 1) * 16 + 15 : -1; break;
    case 0x143: from = l >= 32 ? 31 : -1; break;
    default: std::abort();
    }
    const int got = emu_exchange(src, from < 0 ? l : from);   // every lane takes part in the rendezvous
    const bool enabled = (row_mask >> row) & 1;
    return (enabled && from >= 0) ? got : old;
}
#define __builtin_amdgcn_update_dpp emu_update_dpp
inline int emu_readlane(int v, int src) { return emu_exchange(v, src); }
#define __builtin_amdgcn_readlane emu_readlane
#define __builtin_amdgcn_readfirstlane(x) (x)
// a real lane-0 broadcast inside ONE wavefront (waves of a block may call it different numbers of times)
#define WRSN_WAVE_FIRST_DEFINED
extern int emu_wave_first_slot[4];
inline int wrsn_wave_first(int v) {
    const int w = (int)threadIdx.x >> 6;
    if (((int)threadIdx.x & 63) == 0) emu_wave_first_slot[w] = v;
    emu_wave_barrier();
    const int r = emu_wave_first_slot[w];
    emu_wave_barrier();
    return r;
}
#define __builtin_amdgcn_wave_barrier() emu_wave_barrier()   /* lanes of one wave are fibers here: order their LDS traffic */   /* only used on wave-uniform values */
// the LDS gathers of wrsn_sim.h are inline assembly (eight ds_read back to back): plain loads here
#define WRSN_LDS_GATHER_DEFINED
inline void wrsn_lds_gather8_b32(const int32_t* base, const int (&idx)[8], int (&out)[8]) { for (int k = 0; k < 8; ++k) out[k] = base[idx[k]]; }
inline void wrsn_lds_gather8_b64(const double* base, const int (&idx)[8], double (&out)[8]) { for (int k = 0; k < 8; ++k) out[k] = base[idx[k]]; }
inline int __double2loint(double d) { long long b; std::memcpy(&b, &d, 8); return (int)(b & 0xffffffffll); }
inline int __double2hiint(double d) { long long b; std::memcpy(&b, &d, 8); return (int)(b >> 32); }
inline double __hiloint2double(int hi, int lo) { long long b = ((long long)hi << 32) | (unsigned int)lo; double d; std::memcpy(&d, &b, 8); return d; }

inline float emu_rcpf(float x) { return 1.0f / x; }
#define __builtin_amdgcn_rcpf emu_rcpf
inline float __int_as_float(int i) { float f; std::memcpy(&f, &i, 4); return f; }
inline int __float_as_int(float f) { int i; std::memcpy(&i, &f, 4); return i; }
inline float emu_sqrtf(float x) { return sqrtf(x); }
#define __builtin_amdgcn_sqrtf emu_sqrtf
inline float emu_exp2f(float x) { return exp2f(x); }
#define __builtin_amdgcn_exp2f emu_exp2f

// ---- v_mfma_f32_32x32x2_f32: A lane l -> A[i = l & 31][k = l >> 5], B lane l -> B[k = l >> 5][j = l & 31],
//      D[reg][lane]: col = lane & 31, row = (reg & 3) + 8 * (reg >> 2) + 4 * (lane >> 5)   (one k-ordered fmaf chain)
struct wrsn_v16f { float v[16]; float& operator[](int i) { return v[i]; } const float& operator[](int i) const { return v[i]; } };
#define WRSN_V16F_DEFINED
extern float emu_mfma_a[4][64], emu_mfma_b[4][64];
inline wrsn_v16f emu_mfma_32x32x2(float a, float b, wrsn_v16f c, int, int, int) {
    const int t = (int)threadIdx.x, w = t >> 6, l = t & 63;
    emu_mfma_a[w][l] = a; emu_mfma_b[w][l] = b;
    emu_wave_barrier();
    wrsn_v16f d = c;
    for (int r = 0; r < 16; ++r) {
        const int i = (r & 3) + 8 * (r >> 2) + 4 * (l >> 5), j = l & 31;
        float acc = c[r];
        for (int k = 0; k < 2; ++k) acc = fmaf(emu_mfma_a[w][k * 32 + i], emu_mfma_b[w][k * 32 + j], acc);
        d[r] = acc;
    }
    emu_wave_barrier();
    return d;
}
#define __builtin_amdgcn_mfma_f32_32x32x2f32 emu_mfma_32x32x2

inline int __popcll(unsigned long long v) { return __builtin_popcountll(v); }
inline int atomicAdd(int* p, int v) { int o = *p; *p = o + v; return o; }
inline int atomicMax(int* p, int v) { int o = *p; if (v > o) *p = v; return o; }
inline float __expf(float x) { return expf(x); }

extern int emu_block_order;                  // 0: blocks run in index order, 1: in reverse order (HIP promises no order)
template <typename K, typename... A>
inline void hipLaunchKernelGGL(K kernel, dim3 grid, dim3 block, size_t /*lds*/, hipStream_t, A... args) {
    gridDim.x = grid.x; blockDim.x = block.x;
    for (unsigned b_ = 0; b_ < grid.x; ++b_) {
        const unsigned b = emu_block_order == 1 ? grid.x - 1 - b_ : b_;
        blockIdx.x = b;
        emu_run_block([&]() { kernel(args...); }, block.x);
    }
}
