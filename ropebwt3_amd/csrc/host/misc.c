/* misc.c -- number parsing and timers (restates misc.c:7-16, 116-150 of the reference) */
#include <stdlib.h>
#include <stdint.h>
#include <sys/resource.h>
#include <sys/time.h>
#include "rb3host.h"

int rb3h_verbose = 3;
static double rb3h_t0 = 0;

int64_t rb3h_parse_num(const char *str) /* "7g", "500k", "2.5M" */
{
	char *p;
	double x = strtod(str, &p);
	if (*p == 'G' || *p == 'g') x *= 1e9;
	else if (*p == 'M' || *p == 'm') x *= 1e6;
	else if (*p == 'K' || *p == 'k') x *= 1e3;
	return (int64_t)(x + .499);
}

double rb3h_cputime(void)
{
	struct rusage r;
	getrusage(RUSAGE_SELF, &r);
	return r.ru_utime.tv_sec + r.ru_stime.tv_sec + 1e-6 * (r.ru_utime.tv_usec + r.ru_stime.tv_usec);
}

static double wall(void)
{
	struct timeval tp;
	gettimeofday(&tp, 0);
	return tp.tv_sec + tp.tv_usec * 1e-6;
}

void rb3h_init(void) { rb3h_t0 = wall(); }
double rb3h_realtime(void) { return wall() - rb3h_t0; }

double rb3h_percent_cpu(void)
{
	double rt = rb3h_realtime();
	return rt > 0 ? (rb3h_cputime() + 1e-6) / (rt + 1e-6) : 0;
}

long rb3h_peakrss(void)
{
	struct rusage r;
	getrusage(RUSAGE_SELF, &r);
	return r.ru_maxrss * 1024;
}
