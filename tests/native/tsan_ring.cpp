// ThreadSanitizer run of the host reduction ring (phip_comm_allreduce_host over the "threads" backend, planning-only contexts).
// Built by tests/test_native_sanitizers.py against a -fsanitize=thread build of libpetschip's host code.
#include <petschip.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <thread>
#include <vector>
int main()
{
  const int n = 4, rounds = 20000;
  char key[128] = "tsan-ring";
  std::vector<std::thread> th;
  std::vector<int> bad(n, 0);
  for (int r = 0; r < n; r++)
    th.emplace_back([&, r] {
      phip_ctx ctx; phip_comm c;
      if (phip_init(-1, &ctx) || phip_comm_create(ctx, "threads", key, n, r, &c)) { bad[r] = 1; return; }
      for (int k = 0; k < rounds; k++) {
        double v[3] = {(double)(r + 1), (double)k, 1.0};
        if (phip_comm_allreduce_host(c, v, 3, 0) || v[0] != n * (n + 1) / 2.0 || v[1] != (double)k * n || v[2] != n) { bad[r] = 2; break; }
        double m[1] = {(double)((r * 7 + k) % n)};
        if (phip_comm_allreduce_host(c, m, 1, 1) || m[0] != n - 1) { bad[r] = 3; break; }
      }
      phip_comm_barrier(c);
      phip_comm_destroy(c);
      phip_finalize(ctx);
    });
  for (auto &t : th) t.join();
  for (int r = 0; r < n; r++) if (bad[r]) { printf("rank %d failed (%d)\n", r, bad[r]); return 1; }
  printf("tsan ring ok\n");
  return 0;
}
