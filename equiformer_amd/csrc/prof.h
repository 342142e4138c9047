#pragma once
#include <hip/hip_runtime.h>
// returns a record index (>= 0) when profiling is enabled and `name` passes the filter, else -1
int eqf_prof_begin(const char* name, hipStream_t st, double flops, double bytes);
void eqf_prof_end(int idx, hipStream_t st);
