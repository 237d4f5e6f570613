/* parameter.h -- forwarding header: lets a source file written against the reference's
 * own "parameter.h" (src/parameter.h:9-32) compile unchanged against the MI355X drop-in.
 * Build the caller with -I<repo>/include/sparsebench/compat -DCRS|-DSCS and link
 * libsparsebench_<fmt>.so.  Everything lives in sparsebench.h. */
#ifndef SPARSEBENCH_COMPAT_PARAMETER_H
#define SPARSEBENCH_COMPAT_PARAMETER_H
#include "../sparsebench.h"

#endif
