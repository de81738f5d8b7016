// tests/hipemu/hip_emu.cpp -- fiber scheduler of the CPU emulator (TEST INFRASTRUCTURE ONLY).
#include <hip/hip_runtime.h>
#include <execinfo.h>
#include <signal.h>
#include <unistd.h>

#include <atomic>
#include <condition_variable>
#include <mutex>
#include <thread>
namespace hipemu {
// Every host worker thread emulates whole workgroups on its own: all scheduler state is thread-local, workgroups of one
// launch are handed out through an atomic counter (global-memory atomics of the kernels are real atomics, see hip_runtime.h).
thread_local Tid cur_tid, cur_bid;
thread_local dim3 cur_bdim, cur_gdim;
thread_local uint64_t xchg[1024];
thread_local int nthreads = 0, nalive = 0;

static const size_t STACK = 512 * 1024;
// minimal x86-64 SysV context switch (glibc's swapcontext makes a sigprocmask syscall per switch)
struct Ctx { void *rsp; };
extern "C" void hipemu_switch(Ctx *from, Ctx *to);
asm(R"(
.text
.globl hipemu_switch
.type hipemu_switch,@function
hipemu_switch:
  pushq %rbp
  pushq %rbx
  pushq %r12
  pushq %r13
  pushq %r14
  pushq %r15
  movq %rsp, (%rdi)
  movq (%rsi), %rsp
  popq %r15
  popq %r14
  popq %r13
  popq %r12
  popq %rbx
  popq %rbp
  ret
.size hipemu_switch,.-hipemu_switch
)");
static thread_local Ctx sched_ctx;
static thread_local std::vector<Ctx> lane_ctx;
static thread_local std::vector<char *> stacks;
static thread_local std::vector<char> done;
static thread_local int cur = 0;
static thread_local const std::function<void()> *cur_body = nullptr;

static void segv_handler(int sig) {
  void *bt[64];
  int n = backtrace(bt, 64);
  fprintf(stderr, "hipemu: signal %d in block %u lane %u\n", sig, cur_bid.x, cur_tid.x);
  backtrace_symbols_fd(bt, n, 2);
  fflush(0);
  _exit(139);
}
static bool handler_set = false;
static thread_local void *pending[1024];
void yield_lane(void *site) {
  pending[cur] = site ? site : __builtin_return_address(0);
  hipemu_switch(&lane_ctx[cur], &sched_ctx);
}

// real rendezvous semantics: a workgroup barrier releases when every live thread of the block has arrived,
// a wave rendezvous when every live lane of that wavefront has (waves may run different trip counts in between)
static thread_local int bar_count = 0, bar_gen = 0;
static thread_local int wave_count[16], wave_gen[16], wave_alive[16];
static thread_local int wait_kind[1024], wait_gen[1024];   // 0 runnable, 1 parked at a block barrier, 2 parked at a wave rendezvous
void block_barrier(void *site) {
  int gen = bar_gen;
  if (++bar_count >= nalive) { bar_count = 0; ++bar_gen; return; }
  wait_kind[cur] = 1; wait_gen[cur] = gen;
  while (bar_gen == gen) yield_lane(site);
  wait_kind[cur] = 0;
}
void wave_rendezvous(void *site) {
  int w = cur / 64;
  int gen = wave_gen[w];
  if (++wave_count[w] >= wave_alive[w]) { wave_count[w] = 0; ++wave_gen[w]; return; }
  wait_kind[cur] = 2; wait_gen[cur] = gen;
  while (wave_gen[w] == gen) yield_lane(site);
  wait_kind[cur] = 0;
}

static void lane_entry() {
  (*cur_body)();
  done[cur] = 1;
  --nalive;
  --wave_alive[cur / 64];
  // a thread that exits releases rendezvous points the others are parked at
  if (nalive > 0 && bar_count >= nalive) { bar_count = 0; ++bar_gen; }
  { int w = cur / 64; if (wave_alive[w] > 0 && wave_count[w] >= wave_alive[w]) { wave_count[w] = 0; ++wave_gen[w]; } }
  hipemu_switch(&lane_ctx[cur], &sched_ctx);
  abort();
}

static void run_blocks(dim3 grid, dim3 block, const std::function<void()> &body, std::atomic<long long> &nextBlock) {
  int nt = (int)(block.x * block.y * block.z);
  cur_gdim = grid; cur_bdim = block; cur_body = &body; nthreads = nt;
  if ((int)stacks.size() < nt) { size_t o = stacks.size(); stacks.resize(nt); for (int i = (int)o; i < nt; ++i) stacks[i] = (char *)malloc(STACK); }
  lane_ctx.resize(nt); done.assign(nt, 0);
  const long long nBlocks = (long long)grid.x * grid.y * grid.z;
  for (;;) {
    const long long b = nextBlock.fetch_add(1);
    if (b >= nBlocks) break;
    const unsigned bx = (unsigned)(b % grid.x), by = (unsigned)((b / grid.x) % grid.y), bz = (unsigned)(b / ((long long)grid.x * grid.y));
    cur_bid.x = bx; cur_bid.y = by; cur_bid.z = bz;
    for (int i = 0; i < nt; ++i) {
      // fresh stack: six zeroed callee-saved slots, then lane_entry as the return address (16-byte ABI alignment)
      uintptr_t top = ((uintptr_t)stacks[i] + STACK) & ~(uintptr_t)15;
      void **sp = (void **)(top - 8);
      *--sp = (void *)lane_entry;
      for (int k = 0; k < 6; ++k) *--sp = nullptr;
      lane_ctx[i].rsp = sp;
      done[i] = 0; wait_kind[i] = 0;
    }
    nalive = nt; bar_count = 0;
    for (int w = 0; w < 16; ++w) { wave_count[w] = 0; int lo = w * 64, hi = lo + 64 < nt ? lo + 64 : nt; wave_alive[w] = hi > lo ? hi - lo : 0; }
    while (nalive > 0) {
      for (int i = 0; i < nt; ++i) {
        if (done[i]) continue;
        if (wait_kind[i] == 1 && bar_gen == wait_gen[i]) continue;          // still parked
        if (wait_kind[i] == 2 && wave_gen[i / 64] == wait_gen[i]) continue;
        cur = i;
        cur_tid.x = i % block.x; cur_tid.y = (i / block.x) % block.y; cur_tid.z = i / (block.x * block.y);
        hipemu_switch(&sched_ctx, &lane_ctx[i]);
      }
      // divergence check (HIPEMU_CHECK=1, build with -O0: optimisers duplicate call sites): every live
      // lane of a wave must be parked at the same rendezvous
      static const bool check = getenv("HIPEMU_CHECK") != nullptr;
      for (int w0 = 0; check && w0 < nt; w0 += 64) {
        void *site = nullptr; int first = -1;
        for (int i = w0; i < nt && i < w0 + 64; ++i) {
          if (done[i]) continue;
          if (first < 0) { first = i; site = pending[i]; }
          else if (pending[i] != site) {
            fprintf(stderr, "hipemu: DIVERGENCE in block %u: lane %d parked at %p, lane %d at %p\n", bx, first, site, i, pending[i]);
            void *bt[2] = {site, pending[i]};
            backtrace_symbols_fd(bt, 2, 2);
            abort();
          }
        }
      }
    }
  }
}

// persistent worker pool: fiber stacks and scheduler state live as long as the worker does
namespace {
struct Pool {
  std::mutex mu;
  std::condition_variable cvJob, cvDone;
  std::vector<std::thread> threads;
  long long jobId = 0;
  int wanted = 0, running = 0;
  dim3 grid, block;
  const std::function<void()> *body = nullptr;
  std::atomic<long long> nextBlock{0};
  void worker(int rank) {
    long long seen = 0;
    for (;;) {
      std::unique_lock<std::mutex> lk(mu);
      cvJob.wait(lk, [&] { return jobId != seen; });
      seen = jobId;
      if (rank >= wanted) continue;
      lk.unlock();
      run_blocks(grid, block, *body, nextBlock);
      lk.lock();
      if (--running == 0) cvDone.notify_all();
    }
  }
  void run(int workers, dim3 g, dim3 b, const std::function<void()> &fn) {
    while ((int)threads.size() < workers - 1) { int rank = (int)threads.size(); threads.emplace_back([this, rank] { worker(rank); }); threads.back().detach(); }
    {
      std::lock_guard<std::mutex> lk(mu);
      grid = g; block = b; body = &fn; nextBlock = 0; wanted = workers - 1; running = workers - 1; ++jobId;
    }
    cvJob.notify_all();
    run_blocks(g, b, fn, nextBlock);
    std::unique_lock<std::mutex> lk(mu);
    cvDone.wait(lk, [&] { return running == 0; });
  }
};
Pool &pool() { static Pool *p = new Pool(); return *p; }   // never destroyed: detached workers may outlive static destructors
}  // namespace

void run_grid(dim3 grid, dim3 block, const std::function<void()> &body) {
  if (!handler_set && getenv("HIPEMU_TRACE")) { static char altstack[1 << 16]; stack_t ss; ss.ss_sp = altstack; ss.ss_size = sizeof altstack; ss.ss_flags = 0; sigaltstack(&ss, 0); struct sigaction sa; memset(&sa, 0, sizeof sa); sa.sa_handler = segv_handler; sa.sa_flags = SA_ONSTACK; sigaction(SIGSEGV, &sa, 0); handler_set = true; }
  int nt = (int)(block.x * block.y * block.z);
  if (nt > 1024) { fprintf(stderr, "hipemu: block too large\n"); abort(); }
  static const int maxThreads = [] {
    if (getenv("HIPEMU_CHECK") || getenv("HIPEMU_TRACE")) return 1;
    const char *e = getenv("HIPEMU_THREADS");
    int n = e ? atoi(e) : (int)std::thread::hardware_concurrency();
    return n < 1 ? 1 : (n > 16 ? 16 : n);
  }();
  const long long nBlocks = (long long)grid.x * grid.y * grid.z;
  int workers = (int)(nBlocks < maxThreads ? nBlocks : maxThreads);
  if (workers <= 1) { std::atomic<long long> nextBlock(0); run_blocks(grid, block, body, nextBlock); return; }
  static thread_local bool nested = false;   // launches come from one host thread at a time per pool job; other host threads run serially
  static std::mutex launchMu;
  std::unique_lock<std::mutex> lk(launchMu, std::try_to_lock);
  if (!lk.owns_lock() || nested) { std::atomic<long long> nextBlock(0); run_blocks(grid, block, body, nextBlock); return; }
  nested = true;
  pool().run(workers, grid, block, body);
  nested = false;
}
}  // namespace hipemu
