# Builds the native pieces in-tree (the .so files travel to the GPU box with gpurun):
#   csvplus_amd/lib/libcsvplus_hip.so   HIP kernels + C ABI (gfx950 only)
#   csvplus_amd/lib/libcph_datagen.so   synthetic table generator (CPU, OpenMP)
#   oracle/_build/liboracle.so          CPU restatement of the reference (test infrastructure)
#   tests/cpp/test_host                 C++ host facade tests (reference-style tests)
HIPCC   ?= /opt/rocm/bin/hipcc
ARCH    ?= gfx950
HIPFLAGS = --offload-arch=$(ARCH) -O3 -std=c++17 -fPIC -fvisibility=hidden -Wall -Wno-unused-function
CC      ?= gcc
CXX     ?= g++

LIBDIR  = csvplus_amd/lib
CSRC    = csvplus_amd/csrc
HIP_SRCS = $(CSRC)/capi.hip $(CSRC)/keycodec.hip $(CSRC)/radix_sort.hip $(CSRC)/probe.hip $(CSRC)/chain.hip $(CSRC)/stream_join.hip $(CSRC)/materialize.hip $(CSRC)/csv_ingest.hip $(CSRC)/index_ops.hip $(CSRC)/dist.hip $(CSRC)/calibrate.hip $(CSRC)/small_build.hip $(CSRC)/host_encode.hip $(CSRC)/window_sort.hip $(CSRC)/counted_sort.hip
HIP_OBJS = $(patsubst $(CSRC)/%.hip,$(LIBDIR)/obj/%.o,$(HIP_SRCS))
HIP_HDRS = $(CSRC)/cph_internal.hpp $(CSRC)/device_utils.hpp $(CSRC)/codec_device.hpp $(CSRC)/probe_device.hpp $(CSRC)/hash_device.hpp $(CSRC)/lds_stage.hpp $(CSRC)/host_encode_kernels.hpp include/csvplus_hip.h

all: hip datagen oracle host

hip: $(LIBDIR)/libcsvplus_hip.so
datagen: $(LIBDIR)/libcph_datagen.so
oracle: oracle/_build/liboracle.so oracle/_build/libfaithful.so
host: tests/cpp/test_host tests/cpp/test_host_encode tests/c/abi_demo tests/c/libnccl_standin.so

$(LIBDIR)/obj/%.o: $(CSRC)/%.hip $(HIP_HDRS)
	@mkdir -p $(LIBDIR)/obj
	$(HIPCC) $(HIPFLAGS) -c $< -o $@

$(LIBDIR)/libcsvplus_hip.so: $(HIP_OBJS)
	$(HIPCC) --offload-arch=$(ARCH) -shared -fPIC $(HIP_OBJS) -ldl -o $@

$(LIBDIR)/libcph_datagen.so: $(CSRC)/datagen.c
	@mkdir -p $(LIBDIR)
	$(CC) -O3 -fopenmp -fPIC -shared -fvisibility=hidden -Wall $< -o $@

oracle/_build/liboracle.so: oracle/csvplus_oracle.c
	@mkdir -p oracle/_build
	$(CC) -O2 -fPIC -shared -fvisibility=hidden -Wall $< -o $@

oracle/_build/libfaithful.so: oracle/faithful.cpp
	@mkdir -p oracle/_build
	$(CXX) -O2 -std=c++17 -fopenmp -fPIC -shared -fvisibility=hidden -Wall $< -o $@

tests/cpp/test_host: tests/cpp/test_host.cpp csvplus_amd/host/csvplus.hpp include/csvplus_hip.h $(LIBDIR)/libcsvplus_hip.so
	$(CXX) -O2 -std=c++17 -Wall -Iinclude -Icsvplus_amd/host $< -L$(LIBDIR) -lcsvplus_hip -Wl,-rpath,'$$ORIGIN/../../csvplus_amd/lib' -o $@

# the host-side key encoder's loops and worker pool on their own (CPU only)
tests/cpp/test_host_encode: tests/cpp/test_host_encode.cpp $(CSRC)/host_encode_kernels.hpp
	$(CXX) -O2 -std=c++17 -Wall -pthread $< -o $@

tests/c/abi_demo: tests/c/abi_demo.c include/csvplus_hip.h $(LIBDIR)/libcsvplus_hip.so
	$(CC) -O2 -std=c99 -Wall -Wextra -Iinclude $< -L$(LIBDIR) -lcsvplus_hip -Wl,-rpath,'$$ORIGIN/../../csvplus_amd/lib' -o $@

# NCCL's entry points for thread ranks sharing one GPU (test infrastructure: the RCCL transport of csrc/dist.hip with 2 and 3 ranks)
tests/c/libnccl_standin.so: tests/c/nccl_standin.cpp
	$(HIPCC) -O2 -std=c++17 -fPIC -shared $< -o $@

# ---- sanitizer builds (CPU side only: the checker, the data generator, the host facade's own code) ----------
# tests/test_asan.py runs them; findings abort the run.  The device side has its own canaries: cph_ctx_set_option
# "pool_guard" (tools/gpu_guard.sh runs the whole GPU suite under it).
SAN = -fsanitize=address,undefined -fno-sanitize-recover=undefined -fno-omit-frame-pointer -g -O1
asan: oracle/_build/oracle_fuzz_asan oracle/_build/datagen_asan

oracle/_build/oracle_fuzz_asan: tests/c/oracle_fuzz.c oracle/csvplus_oracle.c
	@mkdir -p oracle/_build
	$(CC) $(SAN) -Wall tests/c/oracle_fuzz.c -o $@

oracle/_build/datagen_asan: tests/c/datagen_check.c $(CSRC)/datagen.c
	@mkdir -p oracle/_build
	$(CC) $(SAN) -fopenmp -Wall tests/c/datagen_check.c -o $@

clean:
	rm -rf $(LIBDIR) oracle/_build tests/cpp/test_host tests/c/abi_demo tests/c/libnccl_standin.so

.PHONY: all hip datagen oracle host asan clean
