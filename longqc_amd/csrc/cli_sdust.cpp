// longqc_amd/csrc/cli_sdust.cpp -- `sdust-mi355x`: same argv and stdout as the reference's `sdust` binary
// (sdust.c:181-222), a drop-in for the path that lq_mask.py:17-23 runs per chunk.  Device: $LQCOV_DEVICE (default 0).
#include "../../include/lqcov.h"
#include <cstdlib>
int main(int argc, char **argv)
{
	const char *d = getenv("LQCOV_DEVICE");
	int rc = lqsdust_main(argc, (const char *const *)argv, nullptr, nullptr, d ? atoi(d) : 0);
	return rc == 0 ? 0 : (rc > 0 ? rc : 3);
}
