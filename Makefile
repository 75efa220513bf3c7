# Build of the MI355X-native Top-K SpMV engine (gfx950 only).
#   make            -> build/libtkspmv.so, bin/approximate-spmv-mi355x-topk, oracle/liboracle.so
#   make ref        -> oracle/_ref/libref_gold.so from the reference's own headers (needs /root/reference)
HIPCC      ?= /opt/rocm/bin/hipcc
CXX        ?= g++
CC         ?= gcc
ARCH       ?= gfx950
REFERENCE  ?= /root/reference

PKG   := approximate-spmv-topk_amd
CSRC  := $(PKG)/csrc
LIB   := $(PKG)/libtkspmv.so
EXE   := bin/approximate-spmv-mi355x-topk
ORACLE:= oracle/liboracle.so
REF   := oracle/_ref/libref_gold.so
REFHOST := oracle/_ref/host_spmv_topk_mi355x

# -ffp-contract=off: products and sums are separately rounded on both sides so the packed-order oracle can be
# matched bit for bit.
HIPFLAGS := -O3 -std=c++17 --offload-arch=$(ARCH) -fPIC -ffp-contract=off -Wall -Wno-unused-result -ldl
CXXFLAGS := -O2 -std=c++17 -fPIC -ffp-contract=off -Wall
CFLAGS   := -O2 -std=c11 -fPIC -ffp-contract=off -Wall -pthread

HOST_SRCS := $(CSRC)/wbscsr.cpp $(CSRC)/wsell.cpp $(CSRC)/host_utils.cpp $(CSRC)/options.cpp $(CSRC)/c_api.cpp
HIP_SRCS  := $(CSRC)/engine.hip $(CSRC)/dist.hip $(CSRC)/device_pack.hip
HDRS      := $(wildcard $(CSRC)/*.hpp) $(wildcard $(CSRC)/kernels/*.hpp) include/tkspmv.h

all: $(LIB) $(EXE) $(ORACLE)

# The compiler's per-kernel resource report (registers, scratch, occupancy) is kept next to the library:
# tests/test_kernel_resources.py fails the build check if a streaming kernel spills or outgrows the 80 registers that let
# two 576-thread workgroups share a CU (a silent 2x slowdown otherwise).
RESOURCES := $(PKG)/kernel_resources.txt
$(LIB): $(HOST_SRCS) $(HIP_SRCS) $(HDRS)
	$(HIPCC) $(HIPFLAGS) -Rpass-analysis=kernel-resource-usage -shared -o $@ $(HIP_SRCS) $(HOST_SRCS) 2> $(RESOURCES).tmp || (cat $(RESOURCES).tmp; rm -f $(RESOURCES).tmp; false)
	@grep -v "kernel-resource-usage\|^ *[0-9]* |\|^ *| *^" $(RESOURCES).tmp >&2 || true
	@grep "remark:" $(RESOURCES).tmp | sed 's/^.*remark: *//; s/ *\[-Rpass-analysis=kernel-resource-usage\]//' > $(RESOURCES); rm -f $(RESOURCES).tmp

$(EXE): $(CSRC)/main_topk.cpp $(LIB) $(HDRS)
	@mkdir -p bin
	$(HIPCC) -O2 -std=c++17 -ffp-contract=off -o $@ $(CSRC)/main_topk.cpp -L$(PKG) -ltkspmv -Wl,-rpath,'$$ORIGIN/../$(PKG)'

$(ORACLE): oracle/oracle.c oracle/hls_model.c oracle/oracle.h
	$(CC) $(CFLAGS) -shared -o $@ oracle/oracle.c oracle/hls_model.c -lm

ref: $(REF) $(REFHOST)
$(REF): oracle/ref_shim.cpp
	@mkdir -p oracle/_ref
	$(CXX) -O2 -std=c++14 -w -fPIC -shared -I$(REFERENCE) -o $@ oracle/ref_shim.cpp

# The reference-side host program of INTEGRATION.md section 2, compiled against the reference's own headers and linked
# with this engine's C ABI (plain g++: no HIP headers on the reference side).
$(REFHOST): oracle/ref_host_mi355x.cpp include/tkspmv.h $(LIB)
	@mkdir -p oracle/_ref
	$(CXX) -O2 -std=c++14 -w -I$(REFERENCE) -Iinclude -o $@ oracle/ref_host_mi355x.cpp -L$(PKG) -ltkspmv -Wl,-rpath,'$$ORIGIN/../../$(PKG)'

clean:
	rm -f $(LIB) $(EXE) $(ORACLE) $(REF) $(REFHOST) $(RESOURCES)

.PHONY: all ref clean
