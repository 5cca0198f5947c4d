// Stable LSD radix sort of 64-bit keys with an optional 32-bit payload (wmf_sort.hip): the ordering primitive behind
// wmf_coo_to_csr (count_mat.T.tocsr(), RecModel/wmf_model.py:128) and the top-n of rank (wmf_model.py:40-47).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

// bytes of the histogram / scan workspace for n elements
size_t wmf_sort_ws_bytes(int64_t n);
// Sorts ascending by the key bits [0, bits) (bits <= 64; higher bits are ignored and must be equal for the order to be total);
// elements with equal keys keep their input order.  keys / vals (vals may be NULL: keys only) are clobbered; keys_alt / vals_alt
// are scratch of the same size.  *in_alt tells where the sorted data ended up (the passes ping-pong).  Enqueues only.
// Returns 0, -2 on a launch failure, -4 when n >= 2^32 (tile offsets and the payload are 32-bit): the code the launchers of
// wmf_csr.hip / wmf_rank.hip pass on and wmf_api.hip reports as WMF_EINVAL ("too many keys"), not as a HIP failure.
int wmf_sort_u64(unsigned long long* keys, unsigned long long* keys_alt, uint32_t* vals, uint32_t* vals_alt, int64_t n, int bits,
                 void* ws, hipStream_t st, bool* in_alt);
