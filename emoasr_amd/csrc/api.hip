// Library-level plumbing: error strings, version, run-time options.
#include <cstdarg>
#include <cstdio>
#include <cstring>
#include "common.h"
#include "../../include/emoasr_hip.h"

static thread_local char g_err[512] = "";

void emo_set_error(const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
}

void emo_gemm_set_tr_read(int v);
void emo_gemm_set_tile(int v);
void emo_gemm_set_tn_group_blocks(int v);
void emo_gemm_set_kb(int v);
void emo_gemm_set_xcd(int v);
void emo_attn_set_tr_read(int v);
void emo_attn_set_fw(int v);

extern "C" const char* emoasr_last_error(void) { return g_err; }
extern "C" int emoasr_version(void) { return 1; }
extern "C" int emoasr_set_option(const char* name, int value) {
  if (strcmp(name, "tr_read") == 0) {
    emo_gemm_set_tr_read(value);
    emo_attn_set_tr_read(value);
    return 0;
  }
  if (strcmp(name, "gemm_tile") == 0) { emo_gemm_set_tile(value); return 0; }
  if (strcmp(name, "tn_group_blocks") == 0) { emo_gemm_set_tn_group_blocks(value); return 0; }
  if (strcmp(name, "gemm_kb") == 0) { emo_gemm_set_kb(value); return 0; }
  if (strcmp(name, "gemm_xcd") == 0) { emo_gemm_set_xcd(value); return 0; }
  if (strcmp(name, "attn_fw") == 0) { emo_attn_set_fw(value); return 0; }
  emo_set_error("unknown option '%s'", name);
  return 1;
}
