// longqc_amd/csrc/cli.cpp -- `minimap2-coverage-mi355x`: same argv as the reference binary
// (minimap2-coverage.c:166-197), table on stdout, log on stderr; a drop-in for the path that
// longQC.py:438-446 hands to LqExec.  Device: $LQCOV_DEVICE (default 0).
#include "../../include/lqcov.h"
#include <cstdlib>
int main(int argc, char **argv)
{
	const char *d = getenv("LQCOV_DEVICE");
	int rc = lqcov_main(argc, (const char *const *)argv, nullptr, nullptr, d ? atoi(d) : 0);
	return rc == 0 ? 0 : (rc > 0 ? rc : 3);
}
