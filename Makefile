# Top-level build: host pipeline (g++), HIP module for gfx950 (hipcc), CPU oracle (gcc).
# No cmake needed; __graft_entry__.build() drives this same file.
ROOT     := $(abspath $(dir $(lastword $(MAKEFILE_LIST))))
PKG      := $(ROOT)/jaderaytracerendering_amd
LIBDIR   := $(PKG)/lib
HIPCC    ?= /opt/rocm/bin/hipcc
CXX      ?= g++
# -ffp-contract=off everywhere: include/jade_fpmath.h pins the evaluation order.
FPFLAGS  := -ffp-contract=off -fno-fast-math
CXXFLAGS := -O2 -g -std=c++17 -fPIC -mfma $(FPFLAGS) -Wall -Wextra -Wno-unused-parameter -I$(ROOT)/include -I$(PKG)/host
# -fno-slp-vectorize: the compiler's own packing of fp32 pairs (v_pk_*_f32) needs aligned register pairs and copies
# (+10...30 VGPRs, measured slower); the packed operations k_trace does use are written by hand (jade_trace.h).
HIPVEC   ?= -fno-slp-vectorize
HIPFLAGS := $(HIPDEFS) -O3 $(HIPVEC) -std=c++17 -fPIC --offload-arch=gfx950 $(FPFLAGS) -fhip-fp32-correctly-rounded-divide-sqrt \
            -fno-gpu-flush-denormals-to-zero -mfma -Wall -Wno-unused-parameter -I$(ROOT)/include -I$(PKG)/csrc

HOST_SRC := $(PKG)/host/scene_build.cpp $(PKG)/host/scene_io.cpp $(PKG)/host/host_capi.cpp
HOST_HDR := $(PKG)/host/jade_host.hpp $(ROOT)/include/jade_host_c.h $(ROOT)/include/jade_rt.h $(ROOT)/include/jade_fpmath.h
HIP_SRC  := $(wildcard $(PKG)/csrc/*.hip)
HIP_HDR  := $(wildcard $(PKG)/csrc/*.h) $(ROOT)/include/jade_rt.h $(ROOT)/include/jade_fpmath.h

all: host hip hipvariants oracle cli

host: $(LIBDIR)/libjade_host.so
hip: $(LIBDIR)/libjade_hip.so
cli: $(LIBDIR)/jade_render
oracle:
	$(MAKE) -C $(ROOT)/oracle

$(LIBDIR)/libjade_host.so: $(HOST_SRC) $(HOST_HDR)
	@mkdir -p $(LIBDIR)
	$(CXX) $(CXXFLAGS) -shared -o $@ $(HOST_SRC)

$(LIBDIR)/libjade_hip.so: $(HIP_SRC) $(HIP_HDR)
	@mkdir -p $(LIBDIR)
	$(HIPCC) $(HIPFLAGS) -shared -o $@ $(HIP_SRC) -ldl

# test-only build: a 4-entry LDS stack forces the global-memory spill path of the
# traversal stack on every scene (tests/test_gpu_parity.py::test_stack_spill_path)
$(LIBDIR)/libjade_hip_stack4.so: $(HIP_SRC) $(HIP_HDR)
	@mkdir -p $(LIBDIR)
	$(HIPCC) -DJADE_LDS_STACK=4 $(HIPFLAGS) -shared -o $@ $(HIP_SRC) -ldl

# test-only build: the jade_debug_* entry points (raw rays with a limit, packets, shadow limits) that the piece-by-piece
# GPU tests call; libjade_hip.so exports exactly what include/jade_rt.h + jade_bvh.h declare
$(LIBDIR)/libjade_hip_debug.so: $(HIP_SRC) $(HIP_HDR)
	@mkdir -p $(LIBDIR)
	$(HIPCC) -DJADE_DEBUG_EXPORTS=1 $(HIPFLAGS) -shared -o $@ $(HIP_SRC) -ldl

hipvariants: $(LIBDIR)/libjade_hip_stack4.so $(LIBDIR)/libjade_hip_debug.so

# development A/B builds (tools/ab_variants.py): make variant NAME=_b256 DEFS="-DJADE_TRACE_BLOCK=256 -DJADE_LDS_TOP_NODES=0"
variant: $(HIP_SRC) $(HIP_HDR)
	@mkdir -p $(LIBDIR)
	$(HIPCC) $(DEFS) $(HIPFLAGS) -shared -o $(LIBDIR)/libjade_hip$(NAME).so $(HIP_SRC) -ldl

$(LIBDIR)/jade_render: $(PKG)/host/jade_render_cli.cpp $(LIBDIR)/libjade_host.so $(HOST_HDR)
	$(CXX) $(CXXFLAGS) -o $@ $(PKG)/host/jade_render_cli.cpp -L$(LIBDIR) -ljade_host -ldl -Wl,-rpath,'$$ORIGIN'

# FETCH_SIZE / TCC_* calibration on k_trace's access pattern (tools/calib/fetch_calib.hip; run under rocprofv3 --pmc)
# VALU issue calibration (tools/calib/valu_calib.hip): wave64 instructions per SIMD per clock, by instruction kind and occupancy
calib: $(LIBDIR)/fetch_calib $(LIBDIR)/valu_calib
$(LIBDIR)/fetch_calib: $(ROOT)/tools/calib/fetch_calib.hip
	@mkdir -p $(LIBDIR)
	$(HIPCC) -O3 -std=c++17 --offload-arch=gfx950 -o $@ $<
$(LIBDIR)/valu_calib: $(ROOT)/tools/calib/valu_calib.hip
	@mkdir -p $(LIBDIR)
	$(HIPCC) -O3 -std=c++17 --offload-arch=gfx950 -o $@ $<

# AddressSanitizer + UBSan over the CPU side (host pipeline, CLI, oracle).  GPU ASan is not
# available on the pool, so this is where memory errors of the non-device code are hunted.
SAN := -fsanitize=address,undefined -fno-omit-frame-pointer -O1 -g
asan-check:
	@mkdir -p $(ROOT)/tests/_build/asan
	$(CXX) $(CXXFLAGS) $(SAN) -shared -o $(ROOT)/tests/_build/asan/libjade_host.so $(HOST_SRC)
	gcc -std=gnu11 -fPIC -ffp-contract=off -mfma $(SAN) -I$(ROOT)/include -shared -o $(ROOT)/tests/_build/asan/libjade_oracle.so $(ROOT)/oracle/jade_oracle.c -lpthread -lm
	$(CXX) $(CXXFLAGS) $(SAN) -o $(ROOT)/tests/_build/asan/jade_render $(PKG)/host/jade_render_cli.cpp -L$(ROOT)/tests/_build/asan -ljade_host -ldl -Wl,-rpath,'$$ORIGIN'
	cd $(ROOT)/tests/_build/asan && for c in tiny tinyjade C1; do \
	  ASAN_OPTIONS=detect_leaks=1 ./jade_render --config $$c --width 48 --height 40 --spp 3 --backend ./libjade_oracle.so --out o_$$c.bmp || exit 1; done
	cd $(ROOT)/tests/_build/asan && ./jade_render --config C2 --width 24 --height 24 --spp 2 --backend ./libjade_oracle.so --out o_C2.ppm
	@echo "asan-check: clean"

clean:
	rm -rf $(LIBDIR)
	$(MAKE) -C $(ROOT)/oracle clean

.PHONY: all host hip hipvariants variant oracle cli clean asan-check calib
