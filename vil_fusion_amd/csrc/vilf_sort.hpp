// vilf_sort.hpp — the library's own stable LSD radix sort of (key, value) pairs in global memory (vilf_sort.hip). Sorts the low `bits` bits of the keys ascending;
// equal keys keep their input order. k_out / v_out receive the result; the inputs are not written. temp: at least vilf_sort_temp_bytes(n, sizeof key) bytes of
// device memory. Everything is enqueued on `st`; returns 0, or a negative value (temp too small / launch error).
#pragma once
#include <hip/hip_runtime.h>
#include <stddef.h>
size_t vilf_sort_temp_bytes(size_t n, size_t key_bytes);
int vilf_sort_pairs_u32(hipStream_t st, void *temp, size_t temp_bytes, const unsigned *k_in, unsigned *k_out, const int *v_in, int *v_out, size_t n, int bits);
int vilf_sort_pairs_u64(hipStream_t st, void *temp, size_t temp_bytes, const unsigned long long *k_in, unsigned long long *k_out, const int *v_in, int *v_out, size_t n, int bits);
