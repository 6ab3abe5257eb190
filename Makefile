# Builds the MI355X-native library (gfx950 only) and the test oracle.
HIPCC ?= /opt/rocm/bin/hipcc
ARCH  ?= gfx950
HIPFLAGS = --offload-arch=$(ARCH) -O3 -std=c++17 -fPIC -Wall -Wno-unused-function -Wno-unused-value
CSRC = dorylus_amd/csrc
OBJS = $(CSRC)/abi.o $(CSRC)/spmm.o $(CSRC)/gemm.o $(CSRC)/elementwise.o
HOSTOBJS = $(patsubst %.cpp,%.o,$(wildcard dorylus_amd/host/*.cpp))
LIB  = dorylus_amd/libdorylus_hip.so

all: $(LIB) oracle

$(CSRC)/%.o: $(CSRC)/%.hip $(CSRC)/ctx.hpp include/dorylus_hip.h
	$(HIPCC) $(HIPFLAGS) -c $< -o $@

dorylus_amd/host/%.o: dorylus_amd/host/%.cpp $(wildcard dorylus_amd/host/*.hpp) $(wildcard include/*.h)
	$(HIPCC) -O3 -std=c++17 -fPIC -fopenmp -Wall -c $< -o $@

$(LIB): $(OBJS) $(HOSTOBJS)
	$(HIPCC) --offload-arch=$(ARCH) -shared -fPIC -fopenmp $(OBJS) $(HOSTOBJS) -L/opt/rocm/lib -lrccl -o $@

oracle:
	$(MAKE) -C oracle

clean:
	rm -f $(OBJS) $(HOSTOBJS) $(LIB); $(MAKE) -C oracle clean

.PHONY: all oracle clean
