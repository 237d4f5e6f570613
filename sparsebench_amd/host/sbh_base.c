/* sbh_base.c -- host utilities of the C side: aligned allocation, wall clock,
 * run-time parameters.  Interfaces follow the reference (src/allocate.h:9,
 * src/timing.h, src/parameter.h:9-22); the code is new.
 */
#define _GNU_SOURCE
#include <errno.h>
#include <stdlib.h>
#include <time.h>

#include "sbhip.h"
#include "sparsebench/sparsebench.h"

/* aligned host memory, fatal on failure: what the library's own host-side arrays come from (the restatement of
 * src/allocate.c:12-36) */
void* sbh_alloc_host(size_t alignment, size_t bytesize)
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

/* allocate() -- the reference's allocation HOOK (src/allocate.h:9, src/allocate.c:12-36).  In the reference everything comes
 * from here; across the drop-in boundary the one caller left is the driver, which takes the vectors of its SpMV mode from it,
 * fills them with host loops and hands them to spMVM (src/main.c:205-215).  So once a device is up, a request of
 * SBH_ALLOCATE_DEVICE_MIN bytes or more comes back as memory that LIVES IN HBM and that the host loop can store to
 * (sb_malloc_host_visible: fine-grained device memory through the PCIe BAR, probed once): spMVM / waxpby / ddot recognise it as
 * a device pointer and run in place -- no staging, nothing over PCIe inside the caller's loop.  Where the probe says no
 * (sb_host_visible_reason), or with SPARSEBENCH_ALLOCATE=host, the request falls back to pinned host memory, then to plain host
 * memory, both staged through HBM by the kernels' wrappers as before.  Host READS of the device-resident kind cross the BAR
 * uncached (slow): a caller that post-processes results on the CPU should copy them out with sb_d2h or use SPARSEBENCH_ALLOCATE=host.
 * sbh_allocate_kind() reports what the last request got; memory from here is never freed by the reference (sbh_allocate_free
 * returns any kind). */
#define SBH_ALLOCATE_DEVICE_MIN ((size_t)64 << 10)
static int g_alloc_kind; /* 0 host, 1 device-resident host-visible, 2 pinned host */
static struct { void* p; int kind; } g_allocs[256];
static int g_nallocs;

void* allocate(size_t alignment, size_t bytesize)
{
  g_alloc_kind = 0;
  if (bytesize >= SBH_ALLOCATE_DEVICE_MIN && sb_is_initialized() && (alignment & (alignment - 1)) == 0 && alignment <= 4096) {
    void* p  = sb_malloc_host_visible(bytesize);
    int kind = 1;
    const char* mode = getenv("SPARSEBENCH_ALLOCATE");
    if (!p && !(mode && strcmp(mode, "host") == 0)) p = sb_malloc_pinned_host(bytesize), kind = 2;
    if (p) {
      g_alloc_kind = kind;
      if (g_nallocs < 256) g_allocs[g_nallocs].p = p, g_allocs[g_nallocs++].kind = kind;
      return p;
    }
  }
  return sbh_alloc_host(alignment, bytesize);
}

int sbh_allocate_kind(void) { return g_alloc_kind; }

void sbh_allocate_free(void* p)
{
  for (int i = 0; i < g_nallocs; i++)
    if (g_allocs[i].p == p) {
      if (g_allocs[i].kind == 1) sb_free(p);
      else sb_free_pinned_host(p);
      g_allocs[i] = g_allocs[--g_nallocs];
      return;
    }
  free(p);
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
