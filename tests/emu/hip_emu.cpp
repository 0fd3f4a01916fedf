// Fiber scheduler of the CPU emulator (see hip_emu.h).  TEST INFRASTRUCTURE ONLY.
#include "hip_emu.h"

#include <vector>
#ifdef EMU_UCONTEXT
#include <ucontext.h>
#endif

namespace emu {

enum State { RUN = 0, WAIT_BLOCK = 1, WAIT_WAVE = 2, DONE = 3 };

struct Fiber {
#ifdef EMU_UCONTEXT
    ucontext_t uc;
#else
    void* sp = nullptr;
#endif
    char* stack = nullptr;
    int state = RUN;
    int lin = 0;  // linear thread id in block
    dim3 tid;
};

dim3 g_blockIdx, g_blockDim, g_gridDim;
Fiber* g_cur = nullptr;

static std::vector<Fiber> g_fibers;
static const std::function<void()>* g_body = nullptr;
static int g_alive = 0, g_block_wait = 0;
static std::vector<int> g_wave_alive, g_wave_wait;
static std::vector<float> g_wave_scratch;
static const size_t kStack = 256 * 1024;

#ifdef EMU_UCONTEXT
static ucontext_t g_main;
static void switch_to_main() { swapcontext(&g_cur->uc, &g_main); }
static void switch_to_fiber(Fiber* f) { g_cur = f; swapcontext(&g_main, &f->uc); }
#else
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
static void* g_main_sp = nullptr;
static void switch_to_main() { emu_switch(&g_cur->sp, g_main_sp); }
static void switch_to_fiber(Fiber* f) { g_cur = f; emu_switch(&g_main_sp, f->sp); }
#endif

static void release_block() {
    for (auto& f : g_fibers)
        if (f.state == WAIT_BLOCK) f.state = RUN;
    g_block_wait = 0;
}
static void release_wave(int w) {
    int lo = w * 64, hi = lo + 64;
    if (hi > (int)g_fibers.size()) hi = (int)g_fibers.size();
    for (int i = lo; i < hi; ++i)
        if (g_fibers[i].state == WAIT_WAVE) g_fibers[i].state = RUN;
    g_wave_wait[w] = 0;
}

static void fiber_main() {
    (*g_body)();
    Fiber* f = g_cur;
    f->state = DONE;
    --g_alive;
    int w = f->lin / 64;
    --g_wave_alive[w];
    // exited threads no longer take part in barriers
    if (g_alive > 0 && g_block_wait == g_alive) release_block();
    if (g_wave_alive[w] > 0 && g_wave_wait[w] == g_wave_alive[w]) release_wave(w);
    switch_to_main();
    fprintf(stderr, "emu: resumed a finished fiber\n");
    abort();
}

#ifdef EMU_UCONTEXT
static void fiber_entry_uc() { fiber_main(); }
#else
extern "C" void emu_fiber_entry() { fiber_main(); }
#endif

const dim3& cur_tid() { return g_cur->tid; }
int cur_lane() { return g_cur->lin & 63; }

void block_barrier() {
    Fiber* f = g_cur;
    f->state = WAIT_BLOCK;
    if (++g_block_wait == g_alive) {
        release_block();
        return;
    }
    switch_to_main();
}

void wave_barrier() {
    Fiber* f = g_cur;
    int w = f->lin / 64;
    f->state = WAIT_WAVE;
    if (++g_wave_wait[w] == g_wave_alive[w]) {
        release_wave(w);
        return;
    }
    switch_to_main();
}

float* wave_scratch() { return g_wave_scratch.data() + (size_t)(g_cur->lin / 64) * 256; }

void launch(dim3 grid, dim3 block, const std::function<void()>& body) {
    const int nthreads = (int)(block.x * block.y * block.z);
    const int nwaves = (nthreads + 63) / 64;
    if (nthreads <= 0 || nthreads > 1024) {
        fprintf(stderr, "emu: bad block size %d\n", nthreads);
        abort();
    }
    g_blockDim = block;
    g_gridDim = grid;
    g_body = &body;
    if ((int)g_fibers.size() < nthreads) {
        size_t old = g_fibers.size();
        g_fibers.resize(nthreads);
        for (size_t i = old; i < g_fibers.size(); ++i) g_fibers[i].stack = (char*)aligned_alloc(64, kStack);
    }
    g_wave_scratch.assign((size_t)nwaves * 256, 0.f);
    for (unsigned bz = 0; bz < grid.z; ++bz)
        for (unsigned by = 0; by < grid.y; ++by)
            for (unsigned bx = 0; bx < grid.x; ++bx) {
                g_blockIdx = dim3(bx, by, bz);
                g_alive = nthreads;
                g_block_wait = 0;
                g_wave_alive.assign(nwaves, 0);
                g_wave_wait.assign(nwaves, 0);
                for (int t = 0; t < nthreads; ++t) {
                    Fiber& f = g_fibers[t];
                    f.state = RUN;
                    f.lin = t;
                    f.tid = dim3(t % block.x, (t / block.x) % block.y, t / (block.x * block.y));
                    ++g_wave_alive[t / 64];
#ifdef EMU_UCONTEXT
                    getcontext(&f.uc);
                    f.uc.uc_stack.ss_sp = f.stack;
                    f.uc.uc_stack.ss_size = kStack;
                    f.uc.uc_link = nullptr;
                    makecontext(&f.uc, fiber_entry_uc, 0);
#else
                    uintptr_t top = ((uintptr_t)(f.stack + kStack)) & ~(uintptr_t)15;
                    void** sp = (void**)top;
                    *(--sp) = nullptr;                   // fake return address of the entry function
                    *(--sp) = (void*)&emu_fiber_entry;   // popped by emu_switch's ret
                    for (int r = 0; r < 6; ++r) *(--sp) = nullptr;  // rbp rbx r12-r15
                    f.sp = sp;
#endif
                }
                // only the first nthreads fibers belong to this launch
                int done = 0;
                while (done < nthreads) {
                    bool progressed = false;
                    done = 0;
                    for (int t = 0; t < nthreads; ++t) {
                        Fiber& f = g_fibers[t];
                        if (f.state == DONE) { ++done; continue; }
                        if (f.state != RUN) continue;
                        progressed = true;
                        switch_to_fiber(&f);
                        if (f.state == DONE) ++done;
                    }
                    if (!progressed && done < nthreads) {
                        fprintf(stderr, "emu: deadlock (divergent barrier?) block=(%u,%u,%u) done=%d/%d block_wait=%d\n", bx, by, bz,
                                done, nthreads, g_block_wait);
                        abort();
                    }
                }
            }
    g_body = nullptr;
    g_cur = nullptr;
}

}  // namespace emu
