/*
 * alloc_host.c -- host-memory flavour of the allocation hooks of shared.h
 * (malloc/memcpy).  Linked into libneutral_host.so, which serves deck parsing
 * and problem set-up; the HBM flavour of the same hooks lives in
 * libneutral_hip.so (csrc/neutral_abi.hip).
 */
#include "shared.h"

#include <string.h>

static void* zalloc(size_t bytes) {
  void* p = calloc(bytes ? bytes : 1, 1);
  if (!p) {
    TERMINATE("Could not allocate %zu bytes of host memory.\n", bytes);
  }
  return p;
}

size_t allocate_data(double** buf, size_t len) {
  *buf = (double*)zalloc(sizeof(double) * len);
  return sizeof(double) * len;
}
size_t allocate_float_data(float** buf, size_t len) {
  *buf = (float*)zalloc(sizeof(float) * len);
  return sizeof(float) * len;
}
size_t allocate_int_data(int** buf, size_t len) {
  *buf = (int*)zalloc(sizeof(int) * len);
  return sizeof(int) * len;
}
size_t allocate_uint64_data(uint64_t** buf, size_t len) {
  *buf = (uint64_t*)zalloc(sizeof(uint64_t) * len);
  return sizeof(uint64_t) * len;
}
void allocate_host_data(double** buf, size_t len) {
  *buf = (double*)zalloc(sizeof(double) * len);
}
void allocate_host_int_data(int** buf, size_t len) {
  *buf = (int*)zalloc(sizeof(int) * len);
}
void deallocate_data(double* buf) { free(buf); }
void deallocate_int_data(int* buf) { free(buf); }
void deallocate_uint64_data(uint64_t* buf) { free(buf); }
void deallocate_host_data(double* buf) { free(buf); }

void copy_buffer(const size_t len, double** src, double** dst, int send) {
  (void)send;
  memcpy(*dst, *src, sizeof(double) * len);
}
void copy_int_buffer(const size_t len, int** src, int** dst, int send) {
  (void)send;
  memcpy(*dst, *src, sizeof(int) * len);
}
void move_host_buffer_to_device(const size_t len, double** src, double** dst) {
  (void)len;
  *dst = *src; /* host "device": adopt the buffer */
  *src = NULL;
}
