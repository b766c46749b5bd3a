# Builds the product without Python: the gfx950 HIP library (C ABI) and the native `mapquik` driver.
# (python __graft_entry__.py does the same and also builds the test-only oracle and input generator.)
HIPCC ?= /opt/rocm/bin/hipcc
CXX   ?= g++
LIBDIR := mapquik_amd/lib
CSRC   := mapquik_amd/csrc

all: $(LIBDIR)/libmapquik_hip.so $(LIBDIR)/mapquik

$(LIBDIR)/libmapquik_hip.so: $(CSRC)/mq_capi.hip $(CSRC)/mq_device.hpp $(CSRC)/mq_seed.hpp include/mapquik_hip.h
	mkdir -p $(LIBDIR)
	$(HIPCC) --offload-arch=gfx950 -O3 -std=c++17 -shared -fPIC -Wno-unused-value -Wno-align-mismatch -o $@ $(CSRC)/mq_capi.hip

$(LIBDIR)/mapquik: $(CSRC)/host/mapquik_main.cc $(CSRC)/host/mapquik_host.hpp $(CSRC)/host/fastx_feeder.hpp include/mapquik_hip.h $(LIBDIR)/libmapquik_hip.so
	$(CXX) -O2 -std=c++17 -Wall -o $@ $(CSRC)/host/mapquik_main.cc -L$(LIBDIR) -lmapquik_hip -lz -lpthread -ldl -Wl,-rpath,'$$ORIGIN' -Wl,-rpath,/opt/rocm/lib

clean:
	rm -rf $(LIBDIR)

.PHONY: all clean
