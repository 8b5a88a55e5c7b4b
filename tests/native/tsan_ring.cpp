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
  /* the threaded set-up passes of phip_mat_create on a planning-only context (SELL-dict arrays: slice lengths, diagonal table by first
     appearance, codes and values): 7-point 56^3, 1.2 M stored entries, 5 threads; the result must not depend on the thread count */
  {
    const int g = 56, m = g * g * g;
    std::vector<int32_t> ai(1, 0), aj;
    std::vector<double>  aa;
    for (int z = 0; z < g; z++) for (int y = 0; y < g; y++) for (int x = 0; x < g; x++) {
      const int r = (z * g + y) * g + x;
      if (z > 0) { aj.push_back(r - g * g); aa.push_back(-1.0); }
      if (y > 0) { aj.push_back(r - g); aa.push_back(-1.0); }
      if (x > 0) { aj.push_back(r - 1); aa.push_back(-1.0); }
      aj.push_back(r); aa.push_back(6.0 + 1e-3 * (r % 7));
      if (x < g - 1) { aj.push_back(r + 1); aa.push_back(-1.0); }
      if (y < g - 1) { aj.push_back(r + g); aa.push_back(-1.0); }
      if (z < g - 1) { aj.push_back(r + g * g); aa.push_back(-1.0); }
      ai.push_back((int32_t)aj.size());
    }
    phip_ctx ctx;
    if (phip_init(-1, &ctx)) { printf("set-up: no planning context\n"); return 1; }
    int32_t nd[2] = {0, 0};
    int64_t pad[2] = {0, 0};
    for (int pass = 0; pass < 2; pass++) {
      phip_mat A;
      int      fmt = 0;
      if (phip_set_setup_threads(pass ? 5 : 1) || phip_mat_create(ctx, m, m, ai.data(), aj.data(), aa.data(), 0, 0, NULL, NULL, &A)) { printf("set-up: create failed: %s\n", phip_last_error()); return 1; }
      if (phip_mat_format_info(A, &fmt, &nd[pass], &pad[pass])) return 1;
      phip_mat_destroy(A);
    }
    phip_set_setup_threads(0);
    phip_finalize(ctx);
    if (nd[0] != 7 || nd[1] != 7 || pad[0] != pad[1] || pad[0] < (int64_t)aj.size()) { printf("set-up: %d / %d diagonals, %lld / %lld padded entries\n", nd[0], nd[1], (long long)pad[0], (long long)pad[1]); return 1; }
  }
  printf("tsan ring ok\n");
  return 0;
}
