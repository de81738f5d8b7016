// tests/hipemu/hip_emu.cpp -- fiber scheduler of the CPU emulator (TEST INFRASTRUCTURE ONLY).
#include <hip/hip_runtime.h>
#include <execinfo.h>
#include <signal.h>
#include <unistd.h>

namespace hipemu {
Tid cur_tid, cur_bid;
dim3 cur_bdim, cur_gdim;
uint64_t xchg[1024];
int nthreads = 0, nalive = 0;

static const size_t STACK = 512 * 1024;
static ucontext_t sched_ctx;
static std::vector<ucontext_t> lane_ctx;
static std::vector<char *> stacks;
static std::vector<char> done;
static int cur = 0;
static const std::function<void()> *cur_body = nullptr;

static void segv_handler(int sig) {
  void *bt[64];
  int n = backtrace(bt, 64);
  fprintf(stderr, "hipemu: signal %d in block %u lane %u\n", sig, cur_bid.x, cur_tid.x);
  backtrace_symbols_fd(bt, n, 2);
  fflush(0);
  _exit(139);
}
static bool handler_set = false;
static void *pending[1024];
void yield_lane(void *site) {
  pending[cur] = site ? site : __builtin_return_address(0);
  swapcontext(&lane_ctx[cur], &sched_ctx);
}

static void lane_entry() {
  (*cur_body)();
  done[cur] = 1;
  --nalive;
  swapcontext(&lane_ctx[cur], &sched_ctx);
}

void run_grid(dim3 grid, dim3 block, const std::function<void()> &body) {
  if (!handler_set && getenv("HIPEMU_TRACE")) { static char altstack[1 << 16]; stack_t ss; ss.ss_sp = altstack; ss.ss_size = sizeof altstack; ss.ss_flags = 0; sigaltstack(&ss, 0); struct sigaction sa; memset(&sa, 0, sizeof sa); sa.sa_handler = segv_handler; sa.sa_flags = SA_ONSTACK; sigaction(SIGSEGV, &sa, 0); handler_set = true; }
  int nt = (int)(block.x * block.y * block.z);
  if (nt > 1024) { fprintf(stderr, "hipemu: block too large\n"); abort(); }
  cur_gdim = grid; cur_bdim = block; cur_body = &body; nthreads = nt;
  if ((int)stacks.size() < nt) { size_t o = stacks.size(); stacks.resize(nt); for (int i = (int)o; i < nt; ++i) stacks[i] = (char *)malloc(STACK); }
  lane_ctx.resize(nt); done.assign(nt, 0);
  for (unsigned bz = 0; bz < grid.z; ++bz) for (unsigned by = 0; by < grid.y; ++by) for (unsigned bx = 0; bx < grid.x; ++bx) {
    cur_bid.x = bx; cur_bid.y = by; cur_bid.z = bz;
    for (int i = 0; i < nt; ++i) {
      getcontext(&lane_ctx[i]);
      lane_ctx[i].uc_stack.ss_sp = stacks[i]; lane_ctx[i].uc_stack.ss_size = STACK; lane_ctx[i].uc_link = &sched_ctx;
      makecontext(&lane_ctx[i], lane_entry, 0);
      done[i] = 0;
    }
    nalive = nt;
    while (nalive > 0) {
      for (int i = 0; i < nt; ++i) {
        if (done[i]) continue;
        cur = i;
        cur_tid.x = i % block.x; cur_tid.y = (i / block.x) % block.y; cur_tid.z = i / (block.x * block.y);
        swapcontext(&sched_ctx, &lane_ctx[i]);
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
}  // namespace hipemu
