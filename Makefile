# Top-level build: the HIP layer (gfx950), the C host side, and the oracle.
# No cmake/ninja needed: hipcc + gcc + make.
HIPCC    ?= /opt/rocm/bin/hipcc
ARCH     ?= gfx950
HIPFLAGS ?= --offload-arch=$(ARCH) -O3 -ffp-contract=off -fPIC -std=c++17 -Wall -Wno-unused-function
LIBDIR    = sparsebench_amd/lib
CSRC      = sparsebench_amd/csrc

all: hip host oracle

hip: $(LIBDIR)/libsbhip.so

$(LIBDIR)/libsbhip.so: $(CSRC)/sbhip.hip $(wildcard $(CSRC)/*.h) include/sbhip.h
	@mkdir -p $(LIBDIR)
	$(HIPCC) $(HIPFLAGS) -shared -Wl,-soname,libsbhip.so -o $@ $(CSRC)/sbhip.hip -ldl

# Lab build: the product plus every measured-slower alternative (levels 1-5 of the compressed mirror as kernels of their
# own, the one-launch vector phase, the lead kernels, hipGraph replay, the two-stream halo overlap, other unroll depths).
# Same file name and soname in a directory of its own, so that the host libraries bind to it when it is loaded first:
#   SBHIP_LIBRARY=$$PWD/sparsebench_amd/lib/lab/libsbhip.so python -m pytest tests -m lab
lab: $(LIBDIR)/lab/libsbhip.so
$(LIBDIR)/lab/libsbhip.so: $(CSRC)/sbhip.hip $(wildcard $(CSRC)/*.h) include/sbhip.h
	@mkdir -p $(LIBDIR)/lab
	$(HIPCC) $(HIPFLAGS) -DSB_LAB=1 -shared -Wl,-soname,libsbhip.so -o $@ $(CSRC)/sbhip.hip -ldl

host: hip
	@if [ -f sparsebench_amd/host/Makefile ]; then $(MAKE) -C sparsebench_amd/host; fi

oracle:
	$(MAKE) -C oracle liboracle.so
	bash oracle/build_ref.sh > /dev/null

clean:
	rm -rf $(LIBDIR) sparsebench_amd/host/build
	$(MAKE) -C oracle clean

.PHONY: all hip host oracle clean lab
