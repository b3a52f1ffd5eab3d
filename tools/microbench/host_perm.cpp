// Micro-benchmark: the host permutation behind the Fiat-Shamir transcript (poseidon_permute_host_fast) and its pieces.
// Build: hipcc -O3 -std=c++17 -x hip --offload-arch=gfx950 host_perm.cpp -o host_perm   (host code only; HIP mode for the shared header)
#include "../../starky_bn254_amd/csrc/poseidon.cuh"
#include <chrono>
#include <cstdio>
int main() {
#if !defined(__HIP_DEVICE_COMPILE__)
  F s[12]; for (int i = 0; i < 12; i++) s[i] = F(i * 1234567ull + 1);
  auto t0 = std::chrono::steady_clock::now();
  for (int i = 0; i < 300000; i++) poseidon_permute_host_fast(s);
  auto t1 = std::chrono::steady_clock::now();
  printf("host fast perm %.3f us (%llx) avx512=%d\n", std::chrono::duration<double>(t1 - t0).count() / 300000 * 1e6, (unsigned long long)s[0].v, (int)host_has_avx512());
  // pieces
  {
    u64 cur = s[0].v, acc = 0;
    t0 = std::chrono::steady_clock::now();
    for (int i = 0; i < 300000; i++)
      for (int k = 0; k < 22; k++) { const u64 t = sbox7(F(cur) + F(POSEIDON_EFF_HOST[12 * k])).v; ph::Acc a; a.lo = acc; ph::mac(a, PFAST_M00, t); cur = ph::fold(a); acc += k; }
    t1 = std::chrono::steady_clock::now();
    printf("bare chain of 22 x (add, S-box, multiply-add, fold) %.3f us (%llx)\n", std::chrono::duration<double>(t1 - t0).count() / 300000 * 1e6, (unsigned long long)cur);
  }
  t0 = std::chrono::steady_clock::now();
  for (int i = 0; i < 300000; i++) { poseidon_mds(s); }
  t1 = std::chrono::steady_clock::now();
  printf("one mds layer %.3f us\n", std::chrono::duration<double>(t1 - t0).count() / 300000 * 1e6);
  t0 = std::chrono::steady_clock::now();
  for (int i = 0; i < 300000; i++) { full_round_sboxes_avx512(reinterpret_cast<u64*>(s), POSEIDON_EFF_HOST); }
  t1 = std::chrono::steady_clock::now();
  printf("12 sboxes avx512 %.3f us\n", std::chrono::duration<double>(t1 - t0).count() / 300000 * 1e6);
#endif
  return 0;
}
