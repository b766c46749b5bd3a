# Builds the product without Python: the gfx950 HIP library (C ABI) and the native `mapquik` driver.
# (python __graft_entry__.py does the same and also builds the test-only oracle and input generator.)
HIPCC ?= /opt/rocm/bin/hipcc
CXX   ?= g++
LIBDIR := mapquik_amd/lib
CSRC   := mapquik_amd/csrc

all: $(LIBDIR)/libmapquik_hip.so $(LIBDIR)/mapquik

# every part of the library's one translation unit (mq_capi.hip includes them all) and both headers: a stale .so would make parity
# and perf numbers describe old kernels (mapquik_amd/build.py globs the same files)
$(LIBDIR)/libmapquik_hip.so: $(CSRC)/mq_capi.hip $(wildcard $(CSRC)/*.hpp) $(wildcard include/*.h)
	mkdir -p $(LIBDIR)
	$(HIPCC) --offload-arch=gfx950 -O3 -std=c++17 -shared -fPIC -Wno-unused-value -Wno-align-mismatch -mllvm -amdgpu-atomic-optimizer-strategy=None -mllvm -pragma-unroll-threshold=65536 -o $@ $(CSRC)/mq_capi.hip

$(LIBDIR)/mapquik: $(CSRC)/host/mapquik_main.cc $(wildcard $(CSRC)/host/*.hpp) $(wildcard include/*.h) $(LIBDIR)/libmapquik_hip.so
	$(CXX) -O2 -std=c++17 -Wall -o $@ $(CSRC)/host/mapquik_main.cc -L$(LIBDIR) -lmapquik_hip -lz -lpthread -ldl -Wl,-rpath,'$$ORIGIN' -Wl,-rpath,/opt/rocm/lib

# Sanitizer builds of the threaded host code (CPU only: no sanitizer runs on the GPU build).  The driver links a host-only stub of
# the C ABI with canned results (tests/cpp/stub_mapquik_hip.cc); tests/test_sanitizers.py builds and runs these.
HOSTSRC := $(CSRC)/host/mapquik_main.cc $(wildcard $(CSRC)/host/*.hpp) $(wildcard include/*.h)
SAN_A := -fsanitize=address,undefined -fno-sanitize-recover=undefined -fno-omit-frame-pointer -g -O1
SAN_T := -fsanitize=thread -fno-omit-frame-pointer -g -O1

asan: $(LIBDIR)/feeder_dump_asan $(LIBDIR)/mapquik_asan
tsan: $(LIBDIR)/feeder_dump_tsan $(LIBDIR)/mapquik_tsan

$(LIBDIR)/feeder_dump_asan: $(CSRC)/host/feeder_dump.cc $(HOSTSRC)
	mkdir -p $(LIBDIR)
	$(CXX) $(SAN_A) -std=c++17 -Wall -o $@ $(CSRC)/host/feeder_dump.cc -lz -lpthread -ldl
$(LIBDIR)/feeder_dump_tsan: $(CSRC)/host/feeder_dump.cc $(HOSTSRC)
	mkdir -p $(LIBDIR)
	$(CXX) $(SAN_T) -std=c++17 -Wall -o $@ $(CSRC)/host/feeder_dump.cc -lz -lpthread -ldl
$(LIBDIR)/mapquik_asan: $(HOSTSRC) tests/cpp/stub_mapquik_hip.cc
	mkdir -p $(LIBDIR)
	$(CXX) $(SAN_A) -std=c++17 -Wall -o $@ $(CSRC)/host/mapquik_main.cc tests/cpp/stub_mapquik_hip.cc -lz -lpthread -ldl
$(LIBDIR)/mapquik_tsan: $(HOSTSRC) tests/cpp/stub_mapquik_hip.cc
	mkdir -p $(LIBDIR)
	$(CXX) $(SAN_T) -std=c++17 -Wall -o $@ $(CSRC)/host/mapquik_main.cc tests/cpp/stub_mapquik_hip.cc -lz -lpthread -ldl

clean:
	rm -rf $(LIBDIR)

.PHONY: all clean asan tsan
