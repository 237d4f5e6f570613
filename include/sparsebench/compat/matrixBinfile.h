/* matrixBinfile.h -- forwarding header: lets a source file written against the reference's
 * own "matrixBinfile.h" (src/matrixBinfile.h:10-22) compile unchanged against the MI355X drop-in.
 * Build the caller with -I<repo>/include/sparsebench/compat -DCRS|-DSCS and link
 * libsparsebench_<fmt>.so.  Everything lives in sparsebench.h. */
#ifndef SPARSEBENCH_COMPAT_MATRIXBINFILE_H
#define SPARSEBENCH_COMPAT_MATRIXBINFILE_H
#include "../sparsebench.h"

#endif
