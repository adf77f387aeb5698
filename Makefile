# Top-level build: everything is built IN-TREE so the .so files travel with the gpurun snapshot.
#   make            -> hip + host + oracle (+ oracle/_ref when /root/reference exists)
#   make hip        -> vision-processor_amd/libvp_hip.so   (hand-written HIP kernels + C ABI, gfx950)
#   make multi      -> vision-processor_amd/libvp_multi.so (multi-GPU gather, the only piece that links RCCL; skipped with a note when librccl is absent)
#   make host       -> vision-processor_amd/libvp_host.so  (C++ host side: synthetic frames, geometry)
#   make facade     -> vision-processor_amd/host/vp_frame_loop (C++ facade demo of main.cpp:262-372) and host/vp_feeds (many feeds, C ABI only)
PKG      := vision-processor_amd
HIPCC    ?= /opt/rocm/bin/hipcc
ARCH     ?= gfx950
# strict IEEE binary32: no contraction, no fast-math
HIPFLAGS := --offload-arch=$(ARCH) -O3 -std=c++17 -fPIC -ffp-contract=off -fno-fast-math -Iinclude -Wall -Wno-unused-function
# host side: strict IEEE binary32 as well (the hypothesis stage is float arithmetic whose order is part of the restatement)
CXXFLAGS := -O2 -std=c++17 -fPIC -Iinclude -Wall -pthread -ffp-contract=off -fno-fast-math

HIP_SRCS := $(filter-out %/vp_multi.hip,$(wildcard $(PKG)/csrc/*.hip))
HIP_HDRS := $(wildcard $(PKG)/csrc/*.h) include/vp_hip.h include/vp_multi.h
HOST_SRCS := $(PKG)/host/synth.cpp $(PKG)/host/geometry.cpp $(wildcard $(PKG)/host/blobs/*.cpp)
FACADE_SRCS := $(PKG)/host/frame_loop.cpp $(PKG)/host/Resources.cpp $(PKG)/host/opencl.cpp

RCCL_LIB := $(firstword $(wildcard /opt/rocm/lib/librccl.so /opt/rocm/lib/librccl.so.1))

.PHONY: all hip multi host oracle facade dbg clean
all: hip multi host oracle facade dbg

hip: $(PKG)/libvp_hip.so
ifneq ($(RCCL_LIB),)
multi: $(PKG)/libvp_multi.so
FEEDS_MULTI_SRC := $(PKG)/host/feeds_multi.cpp
FEEDS_MULTI_LIBS := -lvp_multi
FEEDS_MULTI_DEPS := $(PKG)/libvp_multi.so
else
multi:
	@echo "make multi: librccl not found under /opt/rocm/lib -- libvp_multi.so and vp_feeds --gpus N are skipped (single-GPU library unaffected)"
FEEDS_MULTI_SRC :=
FEEDS_MULTI_LIBS :=
FEEDS_MULTI_DEPS :=
endif
# multi-GPU end (include/vp_multi.h): one process, RCCL all-gather of the per-step records; kept out of libvp_hip.so so that the
# single-GPU library does not depend on librccl
$(PKG)/libvp_multi.so: $(PKG)/csrc/vp_multi.hip $(HIP_HDRS) $(PKG)/libvp_hip.so
	$(HIPCC) $(HIPFLAGS) -I$(PKG)/csrc -shared -o $@ $(PKG)/csrc/vp_multi.hip -L$(PKG) -lvp_hip -L/opt/rocm/lib -lrccl -Wl,-rpath,'$$ORIGIN' -Wl,-rpath,/opt/rocm/lib
$(PKG)/libvp_hip.so: $(HIP_SRCS) $(HIP_HDRS)
	$(HIPCC) $(HIPFLAGS) -shared -o $@ $(HIP_SRCS)

host: $(PKG)/libvp_host.so
$(PKG)/libvp_host.so: $(HOST_SRCS) include/vp_hip.h include/vp_host.h $(wildcard $(PKG)/host/*.h) $(wildcard $(PKG)/host/blobs/*.h)
	g++ $(CXXFLAGS) -shared -o $@ $(HOST_SRCS)

ifneq ($(wildcard $(PKG)/host/frame_loop.cpp),)
facade: $(PKG)/host/vp_frame_loop $(PKG)/host/vp_feeds
else
facade:
endif
$(PKG)/host/vp_frame_loop: $(FACADE_SRCS) $(wildcard $(PKG)/host/*.h) $(PKG)/libvp_hip.so $(PKG)/libvp_host.so
	g++ $(CXXFLAGS) -o $@ $(FACADE_SRCS) -L$(PKG) -lvp_hip -lvp_host -Wl,-rpath,'$$ORIGIN/..' -Wl,-rpath,/opt/rocm/lib

# many feeds on one GPU, C++ on the C ABI alone (vp_frame + read-back + vph_detect_batch, CPU stage overlapped with the next step)
$(PKG)/host/vp_feeds: $(PKG)/host/feeds_main.cpp $(PKG)/host/feeds_multi.cpp $(PKG)/host/feeds.h include/vp_hip.h include/vp_multi.h include/vp_host.h $(PKG)/libvp_hip.so $(FEEDS_MULTI_DEPS) $(PKG)/libvp_host.so
	g++ $(CXXFLAGS) $(if $(FEEDS_MULTI_SRC),-DVP_HAVE_MULTI) -o $@ $(PKG)/host/feeds_main.cpp $(FEEDS_MULTI_SRC) -L$(PKG) $(FEEDS_MULTI_LIBS) -lvp_hip -lvp_host -Wl,-rpath,'$$ORIGIN/..' -Wl,-rpath,/opt/rocm/lib -Wl,-rpath-link,/opt/rocm/lib

oracle:
	$(MAKE) -s -C oracle oracle ref

# device probe behind tests/test_gpu_round3.py::test_dpp_primitives_on_the_device (the gfx950 v_subrev_*_dpp finding)
dbg: tools/dbg/dpp_test
tools/dbg/dpp_test: tools/dbg/dpp_test.hip
	$(HIPCC) --offload-arch=$(ARCH) -O3 -o $@ $<

clean:
	rm -f tools/dbg/dpp_test $(PKG)/libvp_hip.so $(PKG)/libvp_multi.so $(PKG)/libvp_host.so $(PKG)/host/vp_frame_loop $(PKG)/host/vp_feeds
	$(MAKE) -s -C oracle clean
