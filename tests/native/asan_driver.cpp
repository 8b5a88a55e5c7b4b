// Host-logic exerciser for sanitizer runs (AddressSanitizer + UBSan on the PETSc-API mirror, planning-only contexts: no GPU).
// Built and run by tests/test_native_sanitizers.py:
//   g++ -std=c++17 -g -O1 -fsanitize=address,undefined -static-libasan -static-libubsan -Iinclude tests/native/asan_driver.cpp \
//       petsc_amd/csrc/host/petsc_host.cpp -Lpetsc_amd/lib -lpetschip -lpthread -o <tmp>/asan_driver
// Three ranks as threads ("threads" backend): MPIAIJ assembly with under-sized preallocation (rows relocate), every entry inserted by
// rank 0 (stash exchange), MatSetUpMultiply + scatter plan, MatView / MatLoad round trip, MatZeroEntries + re-insertion with the old
// pattern, MatShift (new diagonal entries), error paths.
#include <petschip_host.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <string>
#include <thread>
#include <vector>

#define CHK(call) do { PetscErrorCode _e = (call); if (_e) { fprintf(stderr, "%s:%d: %s -> %d: %s\n", __FILE__, __LINE__, #call, (int)_e, PetscHipLastError()); exit(1); } } while (0)
#define EXPECT_ERR(call, code) do { PetscErrorCode _e = (call); if (_e != (code)) { fprintf(stderr, "%s:%d: %s -> %d, expected %d\n", __FILE__, __LINE__, #call, (int)_e, (int)(code)); exit(1); } } while (0)

static const int G = 7; /* 3-D 7-point stencil on a G^3 grid */

static void rank_main(int size, int rank, const char *dir)
{
  char key[128];
  memset(key, 0, sizeof(key));
  memcpy(key, "asan-key", 8);
  key[8] = (char)size;
  PetscComm comm;
  CHK(PetscHipInitialize(-1, size == 1 ? "self" : "threads", key, size, rank, &comm));
  const PetscInt N = G * G * G;
  Mat A;
  CHK(MatCreate(comm, &A));
  CHK(MatSetSizes(A, PETSC_DECIDE, PETSC_DECIDE, N, N));
  CHK(MatSetType(A, MATAIJHIP));
  if (size > 1) CHK(MatMPIAIJSetPreallocation(A, 2, NULL, 1, NULL)); /* too small on purpose */
  else CHK(MatSeqAIJSetPreallocation(A, 2, NULL));
  PetscInt rs, re;
  CHK(MatGetOwnershipRange(A, &rs, &re));
  auto insert_all = [&](double scale) {
    /* rank 0 inserts every row (the others' rows travel through the stash); columns in descending order to exercise the sorted insert */
    if (rank != 0) return;
    for (PetscInt k = 0; k < G; k++) for (PetscInt j = 0; j < G; j++) for (PetscInt i = 0; i < G; i++) {
      const PetscInt row = i + G * (j + G * k);
      PetscInt    cols[7]; PetscScalar v[7]; int n = 0;
      if (k + 1 < G) { cols[n] = row + G * G; v[n++] = -1.0 * scale; }
      if (j + 1 < G) { cols[n] = row + G; v[n++] = -1.0 * scale; }
      if (i + 1 < G) { cols[n] = row + 1; v[n++] = -1.0 * scale; }
      cols[n] = row; v[n++] = 6.0 * scale;
      if (i > 0) { cols[n] = row - 1; v[n++] = -1.0 * scale; }
      if (j > 0) { cols[n] = row - G; v[n++] = -1.0 * scale; }
      if (k > 0) { cols[n] = row - G * G; v[n++] = -1.0 * scale; }
      CHK(MatSetValues(A, 1, &row, n, cols, v, ADD_VALUES));
      cols[0] = -1; /* negative indices are skipped (aij.c:441) */
      CHK(MatSetValues(A, 1, &row, 1, cols, v, ADD_VALUES));
    }
  };
  insert_all(1.0);
  CHK(MatAssemblyBegin(A, MAT_FINAL_ASSEMBLY));
  CHK(MatAssemblyEnd(A, MAT_FINAL_ASSEMBLY));
  PetscInt m, n, nz, nzr, rmax, cpn; PetscBool cpu;
  CHK(MatHipGetBlockInfo(A, 0, &m, &n, &nz, &nzr, &rmax, &cpu, &cpn));
  if (m != re - rs || rmax > 7 || nzr != m) { fprintf(stderr, "rank %d: block info m=%d nz=%d rmax=%d\n", rank, m, nz, rmax); exit(1); }
  if (size > 1) {
    PetscInt ec, tn, tl, fn;
    CHK(MatHipGetGhostInfo(A, &ec, &tn, &tl, &fn));
    if (ec <= 0 || tn <= 0 || fn <= 0) { fprintf(stderr, "rank %d: no ghosts?\n", rank); exit(1); }
    std::vector<PetscInt> g(ec), tp(tn), ts(tn + 1), ti(tl), fp(fn), fs(fn + 1);
    CHK(MatHipGetGhostArrays(A, g.data(), tp.data(), ts.data(), ti.data(), fp.data(), fs.data()));
    for (PetscInt q = 1; q < ec; q++) if (g[q] <= g[q - 1]) { fprintf(stderr, "garray not sorted\n"); exit(1); }
    if (fs[fn] != ec || ts[tn] != tl) { fprintf(stderr, "plan totals\n"); exit(1); }
  }
  /* file round trip */
  std::string f1 = std::string(dir) + "/a" + std::to_string(size) + ".dat", f2 = std::string(dir) + "/b" + std::to_string(size) + ".dat";
  CHK(MatView(A, f1.c_str(), 0));
  Mat B;
  CHK(MatCreate(comm, &B));
  int64_t off = 0;
  CHK(MatLoad(B, f1.c_str(), &off));
  CHK(MatView(B, f2.c_str(), 0));
  if (rank == 0) {
    FILE *a = fopen(f1.c_str(), "rb"), *b = fopen(f2.c_str(), "rb");
    std::vector<char> ba(1 << 20), bb(1 << 20);
    const size_t na = fread(ba.data(), 1, ba.size(), a), nb = fread(bb.data(), 1, bb.size(), b);
    fclose(a); fclose(b);
    if (na != nb || na != (size_t)off || memcmp(ba.data(), bb.data(), na)) { fprintf(stderr, "MatView/MatLoad round trip differs (%zu %zu %lld)\n", na, nb, (long long)off); exit(1); }
  }
  /* same pattern, new values; then a shift (no new entries here) */
  CHK(MatZeroEntries(A));
  insert_all(2.0);
  CHK(MatAssemblyBegin(A, MAT_FINAL_ASSEMBLY));
  CHK(MatAssemblyEnd(A, MAT_FINAL_ASSEMBLY));
  CHK(MatShift(A, 0.5));
  CHK(MatScale(A, 0.25));
  /* vectors exist on planning contexts, numerical ops do not */
  Vec x, y;
  CHK(MatCreateVecs(A, &x, &y));
  EXPECT_ERR(MatMult(A, x, y), 56);
  EXPECT_ERR(MatMult(A, x, x), 61);
  PetscInt bad = N + 3; PetscScalar one = 1.0;
  if (rank == 0) EXPECT_ERR(MatSetValues(A, 1, &bad, 1, &bad, &one, INSERT_VALUES), 63);
  CHK(VecDestroy(&x)); CHK(VecDestroy(&y));
  CHK(MatDestroy(&B));
  CHK(MatDestroy(&A));
  CHK(PetscHipFinalize(&comm));
}

int main(int argc, char **argv)
{
  const char *dir = argc > 1 ? argv[1] : "/tmp";
  for (int size : {1, 3, 4}) {
    std::vector<std::thread> th;
    for (int r = 0; r < size; r++) th.emplace_back(rank_main, size, r, dir);
    for (auto &t : th) t.join();
  }
  printf("asan driver ok\n");
  return 0;
}
