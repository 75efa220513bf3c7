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

# -ffp-contract=off: products and sums are separately rounded on both sides so the packed-order oracle can be
# matched bit for bit.
HIPFLAGS := -O3 -std=c++17 --offload-arch=$(ARCH) -fPIC -ffp-contract=off -Wall -Wno-unused-result -ldl
CXXFLAGS := -O2 -std=c++17 -fPIC -ffp-contract=off -Wall
CFLAGS   := -O2 -std=c11 -fPIC -ffp-contract=off -Wall -pthread

HOST_SRCS := $(CSRC)/wbscsr.cpp $(CSRC)/wsell.cpp $(CSRC)/host_utils.cpp $(CSRC)/c_api.cpp
HIP_SRCS  := $(CSRC)/engine.hip $(CSRC)/dist.hip $(CSRC)/device_pack.hip
HDRS      := $(wildcard $(CSRC)/*.hpp) $(wildcard $(CSRC)/kernels/*.hpp) include/tkspmv.h

all: $(LIB) $(EXE) $(ORACLE)

$(LIB): $(HOST_SRCS) $(HIP_SRCS) $(HDRS)
	$(HIPCC) $(HIPFLAGS) -shared -o $@ $(HIP_SRCS) $(HOST_SRCS)

$(EXE): $(CSRC)/main_topk.cpp $(LIB) $(HDRS)
	@mkdir -p bin
	$(HIPCC) -O2 -std=c++17 -ffp-contract=off -o $@ $(CSRC)/main_topk.cpp -L$(PKG) -ltkspmv -Wl,-rpath,'$$ORIGIN/../$(PKG)'

$(ORACLE): oracle/oracle.c oracle/oracle.h
	$(CC) $(CFLAGS) -shared -o $@ oracle/oracle.c -lm

ref: $(REF)
$(REF): oracle/ref_shim.cpp
	@mkdir -p oracle/_ref
	$(CXX) -O2 -std=c++14 -w -fPIC -shared -I$(REFERENCE) -o $@ oracle/ref_shim.cpp

clean:
	rm -f $(LIB) $(EXE) $(ORACLE) $(REF)

.PHONY: all ref clean
