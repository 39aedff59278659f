// vf_internal.h -- glue between the host runtime (vf_host.cpp) and the FASTQ reader (vf_fastq.cpp)
#pragma once
#include <stdint.h>

#include <string>

struct vf_ctx;
struct ReadRef { uint64_t off; uint32_t len; };   // a sequence inside a text buffer

int vf_internal_submit(vf_ctx *c, const uint8_t *base, const ReadRef *reads, uint64_t n);
void vf_internal_set_async_error(vf_ctx *c, int rc, const std::string &msg);
uint32_t vf_internal_threads(const vf_ctx *c);
int vf_internal_fail(int code, const std::string &msg);      // sets the calling thread's error text
std::string vf_internal_last_error();
