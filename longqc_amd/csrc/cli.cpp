// longqc_amd/csrc/cli.cpp -- `minimap2-coverage-mi355x`: same argv as the reference binary
// (minimap2-coverage.c:166-197), table on stdout, log on stderr; a drop-in for the path that
// longQC.py:438-446 hands to LqExec.  Device: $LQCOV_DEVICE (default 0).
#include "../../include/lqcov.h"
#include <cstdlib>
#include <cstdio>
#include <unistd.h>
int main(int argc, char **argv)
{
	const char *d = getenv("LQCOV_DEVICE");
	setenv("LQCOV_NO_TEARDOWN", "1", 0);            // this process ends with the call: no block-by-block release of the device memory (lqcov_main)
	int rc = lqcov_main(argc, (const char *const *)argv, nullptr, nullptr, d ? atoi(d) : 0);
	fflush(stdout); fflush(stderr);
	const int code = rc == 0 ? 0 : (rc > 0 ? rc : 3);
	// a profiler or a sanitizer flushes its report from an exit handler: leave the ordinary way then
	if (getenv("ROCPROFILER_REGISTER_FORCE_LOAD") || getenv("ROCP_TOOL_LIBRARIES") || getenv("ROCPROFV3") || getenv("LD_PRELOAD") || getenv("ASAN_OPTIONS") || getenv("GCOV_PREFIX")) exit(code);
	_exit(code);                                    // (the driver reclaims everything at once)
}
