// TEST INFRASTRUCTURE ONLY -- fiber scheduler of the lockstep wavefront emulator (see include/hip/hip_runtime.h).
#include <hip/hip_runtime.h>

#include <cstdio>
#include <vector>

emu_idx threadIdx, blockIdx, blockDim, gridDim;
alignas(16) double smem[160 * 1024 / 8];
unsigned char emu_slots[2][256][16];
int emu_parity = 0;
int emu_block_order = 0;
int emu_wave_first_slot[4];
extern "C" void emu_set_block_order(int o) { emu_block_order = o; }
float emu_mfma_a[4][64], emu_mfma_b[4][64];

extern "C" void emu_switch(void** save_sp, void* load_sp);
asm(R"(
.text
.globl emu_switch
.type emu_switch,@function
emu_switch:
    pushq %rbp
    pushq %rbx
    pushq %r12
    pushq %r13
    pushq %r14
    pushq %r15
    movq %rsp, (%rdi)
    movq %rsi, %rsp
    popq %r15
    popq %r14
    popq %r13
    popq %r12
    popq %rbx
    popq %rbp
    ret
.size emu_switch,.-emu_switch
)");

namespace {
constexpr size_t kStack = 256 * 1024;
constexpr unsigned kMaxThreads = 256;
void* g_main_sp;
void* g_fiber_sp[kMaxThreads];
std::vector<char> g_stacks;
const std::function<void()>* g_body;
bool g_done[kMaxThreads];
int g_waiting[kMaxThreads];                     // 0 running, 1 at a block-wide rendezvous, 2 at a rendezvous of its wavefront
unsigned g_cur;

void fiber_entry() {
    (*g_body)();
    g_done[g_cur] = true;
    emu_switch(&g_fiber_sp[g_cur], g_main_sp);
    std::abort();
}
}  // namespace

void emu_barrier() {
    g_waiting[g_cur] = 1;
    unsigned me = g_cur;
    emu_switch(&g_fiber_sp[me], g_main_sp);      // resumed once every live fiber has arrived
    threadIdx.x = me;
}

// rendezvous of the 64 lanes of the caller's wavefront only (wave-level instructions such as MFMA: the waves of a
// workgroup may issue different numbers of them between two __syncthreads)
void emu_wave_barrier() {
    g_waiting[g_cur] = 2;
    unsigned me = g_cur;
    emu_switch(&g_fiber_sp[me], g_main_sp);
    threadIdx.x = me;
}

void emu_run_block(const std::function<void()>& body, unsigned nthreads) {
    if (nthreads > kMaxThreads) std::abort();
    if (g_stacks.size() < kStack * kMaxThreads) g_stacks.resize(kStack * kMaxThreads);
    g_body = &body;
    for (unsigned t = 0; t < nthreads; ++t) {
        char* top = g_stacks.data() + kStack * (t + 1);
        top = (char*)((uintptr_t)top & ~(uintptr_t)15);
        void** sp = (void**)top;
        *--sp = nullptr;                          // fake return address: entry sees rsp % 16 == 8
        *--sp = (void*)&fiber_entry;
        for (int i = 0; i < 6; ++i) *--sp = nullptr;
        g_fiber_sp[t] = sp;
        g_done[t] = false; g_waiting[t] = false;
    }
    for (unsigned t = 0; t < nthreads; ++t) g_waiting[t] = 0;
    for (;;) {
        bool progressed = false;
        for (unsigned t = 0; t < nthreads; ++t) {
            if (g_done[t] || g_waiting[t]) continue;
            g_cur = t; threadIdx.x = t;
            emu_switch(&g_main_sp, g_fiber_sp[t]);
            progressed = true;
        }
        unsigned live = 0, at_block = 0;
        for (unsigned t = 0; t < nthreads; ++t) if (!g_done[t]) { ++live; if (g_waiting[t] == 1) ++at_block; }
        if (live == 0) break;
        bool released = false;
        for (unsigned w0 = 0; w0 < nthreads; w0 += 64) {      // wavefront rendezvous first
            unsigned lw = 0, ww = 0;
            for (unsigned t = w0; t < w0 + 64 && t < nthreads; ++t) if (!g_done[t]) { ++lw; if (g_waiting[t] == 2) ++ww; }
            if (lw > 0 && ww == lw) { for (unsigned t = w0; t < w0 + 64 && t < nthreads; ++t) if (!g_done[t]) g_waiting[t] = 0; released = true; }
        }
        if (!released && at_block == live) {
            for (unsigned t = 0; t < nthreads; ++t) g_waiting[t] = 0;
            emu_parity ^= 1; released = true;
        }
        if (!released && !progressed) { std::fprintf(stderr, "emu: divergent rendezvous (%u of %u fibers at the block barrier)\n", at_block, live); std::abort(); }
    }
}
