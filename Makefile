# Builds libsmr_hip.so (gfx950 only) in-tree and the test-only oracle libraries.
HIPCC ?= /opt/rocm/bin/hipcc
ARCH ?= gfx950
CSRC := smrender_amd/csrc
HIPFLAGS := -O3 -std=c++17 -fPIC --offload-arch=$(ARCH) -ffp-contract=off -Iinclude -I$(CSRC) -Wall -Wno-unused-function -Wno-unused-value -Wno-unused-result
SRCS := $(wildcard $(CSRC)/*.hip) $(wildcard $(CSRC)/*.cpp)
OBJS := $(patsubst %,build/%.o,$(notdir $(SRCS)))

all: smrender_amd/libsmr_hip.so oracle

build/%.hip.o: $(CSRC)/%.hip $(CSRC)/smr_dev.h $(CSRC)/smr_host.h $(CSRC)/raster_cross.inc include/smr_hip.h
	@mkdir -p build
	$(HIPCC) $(HIPFLAGS) -c $< -o $@

build/%.cpp.o: $(CSRC)/%.cpp include/smr_hip.h
	@mkdir -p build
	$(HIPCC) $(HIPFLAGS) -x hip -c $< -o $@

smrender_amd/libsmr_hip.so: $(OBJS)
	$(HIPCC) -shared -fPIC --offload-arch=$(ARCH) -o $@ $(OBJS) -ldl -lz -lpthread

oracle:
	$(MAKE) -C oracle all

clean:
	rm -rf build smrender_amd/libsmr_hip.so
	$(MAKE) -C oracle clean

.PHONY: all oracle clean prof

# diagnostic build with s_memtime stamps around the sections of k_raster (never the shipped library)
prof: tools/libsmr_hip_prof.so
tools/libsmr_hip_prof.so: $(SRCS) $(CSRC)/smr_dev.h $(CSRC)/smr_host.h include/smr_hip.h
	@mkdir -p build
	$(HIPCC) $(HIPFLAGS) -DSMR_SECTION_PROFILE -x hip -shared -o $@ $(SRCS) -ldl -lz -lpthread
