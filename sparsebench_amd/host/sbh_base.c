/* sbh_base.c -- host utilities of the C side: aligned allocation, wall clock,
 * run-time parameters.  Interfaces follow the reference (src/allocate.h:9,
 * src/timing.h, src/parameter.h:9-22); the code is new.
 */
#define _GNU_SOURCE
#include <errno.h>
#include <stdlib.h>
#include <time.h>

#include "sparsebench/sparsebench.h"

/* src/allocate.c:12-36: aligned host memory, fatal on failure */
void* allocate(size_t alignment, size_t bytesize)
{
  void* p = NULL;
  int rc  = posix_memalign(&p, alignment, bytesize ? bytesize : alignment);
  if (rc == EINVAL) {
    fprintf(stderr, "Error: Alignment parameter is not a power of two\n");
    exit(EXIT_FAILURE);
  }
  if (rc == ENOMEM || p == NULL) {
    fprintf(stderr, "Error: Insufficient memory to fulfill the request\n");
    exit(EXIT_FAILURE);
  }
  return p;
}

/* src/timing.c:8-13: CLOCK_MONOTONIC seconds */
double getTimeStamp(void)
{
  struct timespec t;
  clock_gettime(CLOCK_MONOTONIC, &t);
  return (double)t.tv_sec + 1.e-9 * (double)t.tv_nsec;
}

/* src/timing.c:15-28: smallest observable step of getTimeStamp() */
double getTimeResolution(void)
{
  struct timespec t;
  if (clock_getres(CLOCK_MONOTONIC, &t) == 0) return (double)t.tv_sec + 1.e-9 * (double)t.tv_nsec;
  double a = getTimeStamp(), b;
  while ((b = getTimeStamp()) == a) {}
  return b - a;
}

/* src/parameter.c:12-20: defaults generate,100,100,100,150,0.0 */
void initParameter(Parameter* p)
{
  p->filename = "generate";
  p->nx = p->ny = p->nz = 100;
  p->itermax            = 150;
  p->eps                = 0.0;
}

/* src/parameter.c:22-62: "key value #comment" lines; a string value needs a blank
 * after it (hpcg.par:5) because tokens are split on ' ' only */
void readParameter(Parameter* p, const char* path)
{
  FILE* f = fopen(path, "r");
  if (!f) {
    fprintf(stderr, "Could not open parameter file: %s\n", path);
    exit(EXIT_FAILURE);
  }
  char line[4096];
  while (fgets(line, sizeof line, f)) {
    char* hash = strchr(line, '#');
    if (hash) *hash = '\0';
    char* save = NULL;
    char* key  = strtok_r(line, " \t\r\n", &save);
    char* val  = key ? strtok_r(NULL, " \t\r\n", &save) : NULL;
    if (!key || !val) continue;
    if (strcmp(key, "filename") == 0) p->filename = strdup(val);
    else if (strcmp(key, "nx") == 0) p->nx = atoi(val);
    else if (strcmp(key, "ny") == 0) p->ny = atoi(val);
    else if (strcmp(key, "nz") == 0) p->nz = atoi(val);
    else if (strcmp(key, "itermax") == 0) p->itermax = atoi(val);
    else if (strcmp(key, "eps") == 0) p->eps = atof(val);
  }
  fclose(f);
}

void printParameter(Parameter* p)
{
  printf("Parameters\n");
  printf("\tfilename: %s\n", p->filename);
  printf("\tnx, ny, nz: %d, %d, %d\n", p->nx, p->ny, p->nz);
  printf("\titermax: %d\n", p->itermax);
  printf("\teps: %e\n", p->eps);
}
