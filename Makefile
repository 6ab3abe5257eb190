# Builds the MI355X-native library (gfx950 only) and the test oracle.
HIPCC ?= /opt/rocm/bin/hipcc
ARCH  ?= gfx950
HIPFLAGS = --offload-arch=$(ARCH) -O3 -std=c++17 -fPIC -Wall -Wno-unused-function -Wno-unused-value $(EXTRA)
CSRC = dorylus_amd/csrc
OBJS = $(CSRC)/abi_context.o $(CSRC)/abi_stages.o $(CSRC)/abi_comm.o $(CSRC)/spmm.o $(CSRC)/spmm_blocked.o $(CSRC)/gemm.o $(CSRC)/elementwise.o $(CSRC)/gat_mh.o $(CSRC)/gat_mh_blocked.o $(CSRC)/gat_mh_sweep.o
HOSTOBJS = $(patsubst %.cpp,%.o,$(filter-out %_main.cpp,$(wildcard dorylus_amd/host/*.cpp)))
GRAPHSERVER = dorylus_amd/graphserver
INPUTS = dorylus_amd/dory-inputs
LIB  = dorylus_amd/libdorylus_hip.so

all: $(LIB) $(GRAPHSERVER) $(INPUTS) oracle

$(CSRC)/%.o: $(CSRC)/%.hip $(CSRC)/ctx.hpp $(CSRC)/spmm_common.hpp $(CSRC)/sweep_core.hpp $(CSRC)/abi_internal.hpp $(CSRC)/gat_mh.hpp include/dorylus_hip.h
	$(HIPCC) $(HIPFLAGS) -c $< -o $@

dorylus_amd/host/%.o: dorylus_amd/host/%.cpp $(wildcard dorylus_amd/host/*.hpp) $(wildcard include/*.h)
	$(HIPCC) -O3 -std=c++17 -fPIC -fopenmp -Wall -c $< -o $@

$(LIB): $(OBJS) $(HOSTOBJS)
	$(HIPCC) --offload-arch=$(ARCH) -shared -fPIC -fopenmp $(OBJS) $(HOSTOBJS) -L/opt/rocm/lib -lrccl -o $@

# the reference's `graphserver` command line on the hip backend (run/run-onnode:154-179)
$(GRAPHSERVER): dorylus_amd/host/graphserver_main.cpp $(LIB)
	g++ -O2 -std=c++17 -Iinclude dorylus_amd/host/graphserver_main.cpp -Ldorylus_amd -ldorylus_hip -Wl,-rpath,'$$ORIGIN' -o $@

# inputs/ tool equivalents (graphtobinary, featurestobinary, labelstobinary, partitioner)
$(INPUTS): dorylus_amd/host/inputs_main.cpp
	g++ -O2 -std=c++17 dorylus_amd/host/inputs_main.cpp -o $@

oracle:
	$(MAKE) -C oracle

clean:
	rm -f $(OBJS) $(HOSTOBJS) $(LIB) $(GRAPHSERVER) $(INPUTS); $(MAKE) -C oracle clean

.PHONY: all oracle clean
