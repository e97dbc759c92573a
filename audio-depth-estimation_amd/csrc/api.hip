// Error plumbing and version of libadn (C ABI, see include/adn.h).
#include <stdarg.h>
#include <string.h>

#include "adn_common.h"

static thread_local char g_err[512] = "";

void adn_set_error(const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
}

extern "C" const char* adn_last_error(void) { return g_err; }
// 2: AdnWgradDesc grew (sq_partials); adn_wgrad_sq_count, adn_grad_norm_ranges, adn_loss_finish_dz added
extern "C" int adn_version(void) { return 2; }
