// hip_emu.cpp -- fiber scheduler for the CPU-side SIMT emulator (test infrastructure only; see
// hip_emu.h).  Never linked into the product library.
#include "hip_emu.h"
#include <sys/mman.h>
#include <mutex>

namespace emu {
State g;
uint3_emu threadIdx, blockIdx;
dim3 blockDim, gridDim;

static const size_t kStack = 256 * 1024;

#ifdef EMU_FAST_SWITCH
// save the callee-saved registers on the current stack, park its stack pointer in *save, continue on `load`
extern "C" void emu_switch(void** save, void* load);
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
.size emu_switch, .-emu_switch
)");
#define EMU_TO_SCHED(f) emu_switch(&(f).sp, g.sched)
#define EMU_TO_FIBER(f) emu_switch(&g.sched, (f).sp)
#else
#define EMU_TO_SCHED(f) swapcontext(&(f).ctx, &g.sched)
#define EMU_TO_FIBER(f) swapcontext(&g.sched, &(f).ctx)
#endif

static void fiber_entry() {
    g.body();
    Fiber& f = g.fibers[g.cur];
    f.done = true;
    // a finished lane no longer takes part in barriers / wave exchanges
    g.live_threads--;
    WaveSync& w = g.waves[f.tid >> 6];
    w.live--;
    // if the peers were only waiting for this lane, release them
    if (w.live > 0 && w.arrived == w.live) { w.arrived = 0; w.gen++; }
    if (g.live_threads > 0 && g.bar_arrived == g.live_threads) { g.bar_arrived = 0; g.bar_gen++; }
    EMU_TO_SCHED(f);
    abort();   // a finished fiber is never resumed
}

void yield_wait(volatile int* var, int val) {
    Fiber& f = g.fibers[g.cur];
    f.wait_var = var;
    f.wait_val = val;
    EMU_TO_SCHED(f);
    // resumed: restore ids (scheduler sets them)
}

static void run_block(dim3 block, unsigned bx, size_t shmem) {
    unsigned n = block.x;
    g.fibers.assign(n, Fiber());
    g.waves.assign((n + 63) / 64, WaveSync());
    g.bar_arrived = 0;
    g.bar_gen = 0;
    g.live_threads = (int)n;
    // exactly-sized heap block so ASan sees LDS overruns
    unsigned char* smem = shmem ? (unsigned char*)aligned_alloc(16, (shmem + 15) & ~(size_t)15) : nullptr;
    if (smem) memset(smem, 0xA5, shmem);  // LDS is garbage at kernel start on hardware
    g.dyn_smem = smem;
    static std::vector<void*> stacks;
    while (stacks.size() < n) {
        void* s = mmap(nullptr, kStack, PROT_READ | PROT_WRITE, MAP_PRIVATE | MAP_ANONYMOUS | MAP_NORESERVE, -1, 0);
        if (s == MAP_FAILED) { perror("mmap"); abort(); }
        stacks.push_back(s);
    }
    for (unsigned t = 0; t < n; ++t) {
        Fiber& f = g.fibers[t];
        f.tid = t;
        f.stack = stacks[t];
        g.waves[t >> 6].live++;
#ifdef EMU_FAST_SWITCH
        // first switch "returns" into fiber_entry: six zeroed callee-saved slots, the entry address in a 16-byte
        // aligned slot (so the function starts with the stack alignment of a normal call), a null return address
        void** top = (void**)((char*)f.stack + kStack);
        top[-1] = nullptr;
        top[-2] = (void*)fiber_entry;
        for (int k = 3; k <= 8; ++k) top[-k] = nullptr;
        f.sp = (void*)(top - 8);
#else
        getcontext(&f.ctx);
        f.ctx.uc_stack.ss_sp = f.stack;
        f.ctx.uc_stack.ss_size = kStack;
        f.ctx.uc_link = nullptr;
        makecontext(&f.ctx, (void (*)())fiber_entry, 0);
#endif
    }
    blockIdx.x = bx; blockIdx.y = 0; blockIdx.z = 0;
    unsigned remaining = n;
    unsigned long idle_passes = 0, spin_passes = 0;
    while (remaining) {
        bool progressed = false;
        bool real_progress = false;
        g.pass_counter = g.pass_counter + 1;
        for (unsigned t = 0; t < n; ++t) {
            Fiber& f = g.fibers[t];
            if (f.done) continue;
            if (f.wait_var) {
                if (*f.wait_var == f.wait_val) continue;  // still parked
                f.wait_var = nullptr;
            }
            g.cur = t;
            threadIdx.x = t; threadIdx.y = 0; threadIdx.z = 0;
            g.last_yield_was_spin = false;
            EMU_TO_FIBER(f);
            progressed = true;
            if (!g.last_yield_was_spin) real_progress = true;
            if (f.done) remaining--;
        }
        if (!progressed) {
            if (++idle_passes > 2) {
                fprintf(stderr, "[hip_emu] DEADLOCK in block %u: %u threads parked (barrier arrived=%d live=%d)\n",
                        bx, remaining, g.bar_arrived, g.live_threads);
                for (size_t w = 0; w < g.waves.size(); ++w)
                    fprintf(stderr, "  wave %zu: arrived=%d live=%d\n", w, g.waves[w].arrived, g.waves[w].live);
                abort();
            }
        } else {
            idle_passes = 0;
        }
        if (progressed && !real_progress) {
            if (++spin_passes > 2000000ul) {
                fprintf(stderr, "[hip_emu] LIVELOCK in block %u: %u threads only spin-waiting\n", bx, remaining);
                abort();
            }
        } else {
            spin_passes = 0;
        }
    }
    free(smem);
    g.dyn_smem = nullptr;
}

void launch(dim3 grid, dim3 block, size_t shmem, std::function<void()> body) {
    // one kernel at a time: the scheduler state, the built-in index variables and the kernels' static LDS are process
    // globals, while the stream ABI runs device work of different host threads side by side
    static std::mutex one_launch;
    std::lock_guard<std::mutex> lk(one_launch);
    g.body = body;
    blockDim = block;
    gridDim = grid;
    for (unsigned bx = 0; bx < grid.x; ++bx) run_block(block, bx, shmem);
}
}  // namespace emu
