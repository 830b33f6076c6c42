// hdu_host.h -- host-side helpers shared by the C-ABI translation units.
#pragma once
#include "../../include/hdu.h"
#include "hdu_platform.h"

int hdu_set_error(int code, const char* msg);
int hdu_check_launch(const char* what);

static inline unsigned hdu_grid_1d(long long work_items, int per_block, unsigned cap) {
  long long b = (work_items + per_block - 1) / per_block;
  if (b < 1) b = 1;
  if (b > (long long)cap) b = cap;
  return (unsigned)b;
}

extern int g_tuning[32];   // hdu_set_tuning values (conv_igemm.hip)
