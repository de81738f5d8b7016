// tests/hipemu/hip_emu.cpp -- fiber scheduler of the CPU emulator (TEST INFRASTRUCTURE ONLY).
#include <hip/hip_runtime.h>

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

void yield_lane() { swapcontext(&lane_ctx[cur], &sched_ctx); }

static void lane_entry() {
  (*cur_body)();
  done[cur] = 1;
  --nalive;
  swapcontext(&lane_ctx[cur], &sched_ctx);
}

void run_grid(dim3 grid, dim3 block, const std::function<void()> &body) {
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
    }
  }
}
}  // namespace hipemu
