// TEST INFRASTRUCTURE ONLY -- runtime of the host SIMT emulator (see include/hip/hip_runtime.h).
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstring>

namespace ha { alignas(16) float smem[40960]; }   // 160 KiB "LDS", one block resident at a time

namespace simt_emu {
thread_local dim3 t_threadIdx, t_blockIdx, t_blockDim, t_gridDim;
thread_local BlockCtx* t_ctx = nullptr;

void launch(dim3 grid, dim3 block, const std::function<void()>& body) {
  const unsigned nthreads = block.x * block.y * block.z;
  const unsigned nwaves = (nthreads + 63) / 64;
  if (block.y != 1 || block.z != 1 || nthreads % 64 != 0) {
    std::fprintf(stderr, "simt_emu: only 1-D blocks of whole waves are supported\n");
    std::abort();
  }
  // LDS is not cleared between kernels on the hardware, and a kernel that reads a word it never wrote sees whatever the previous tenant of the
  // CU left there (round 5: NaN gradients on one box, green on another).  Every emulated launch therefore starts on NaN-filled "LDS".
  {
    const unsigned nanbits = 0x7fc00000u;
    float nanv;
    std::memcpy(&nanv, &nanbits, 4);
    std::fill(ha::smem, ha::smem + sizeof(ha::smem) / sizeof(float), nanv);
  }
  BlockCtx ctx;
  std::barrier<> block_bar(nthreads);
  ctx.block_bar = &block_bar;
  for (unsigned w = 0; w < nwaves; ++w) {
    ctx.wave_bar.emplace_back(new std::barrier<>(64));
    ctx.xch.emplace_back(128, 0);
  }
  const unsigned long long nblocks = (unsigned long long)grid.x * grid.y * grid.z;
  std::vector<std::thread> threads;
  threads.reserve(nthreads);
  for (unsigned t = 0; t < nthreads; ++t) {
    threads.emplace_back([&, t]() {
      t_ctx = &ctx;
      t_blockDim = block;
      t_gridDim = grid;
      t_threadIdx = dim3(t, 0, 0);
      for (unsigned long long b = 0; b < nblocks; ++b) {
        t_blockIdx = dim3((unsigned)(b % grid.x), (unsigned)((b / grid.x) % grid.y), (unsigned)(b / ((unsigned long long)grid.x * grid.y)));
        body();
        // a thread that returned early from the kernel must not leave the others stuck: kernels under test
        // only return early wave-uniformly and never before a later barrier, so a plain block barrier here
        // keeps blocks from overlapping in the shared LDS array.
        ctx.block_bar->arrive_and_wait();
      }
    });
  }
  for (auto& th : threads) th.join();
}
}  // namespace simt_emu
