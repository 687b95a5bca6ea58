# Top-level build: the product library (HIP, gfx950 only), the host-side library, the oracle.
HIPCC    ?= /opt/rocm/bin/hipcc
ARCH     ?= gfx950
PKG      := vulkan_raytracing_amd
CSRC     := $(PKG)/csrc
# -ffp-contract=off: the kernels' arithmetic is the canonical sequence of DESIGN.md; only explicit
# __builtin_fmaf fuses.  Division and sqrt stay IEEE-correct (hipcc default).
# -fno-slp-vectorize: hipcc otherwise packs adjacent f32 ops into v_pk_*_f32, which issue slower than the
# scalar forms on gfx950 (MI355X_MICROARCH.md 'price of one filler'; measured on the traversal kernels).
HIPFLAGS := -O3 -std=c++17 --offload-arch=$(ARCH) -ffp-contract=off -fno-slp-vectorize -fPIC -Wall -Wno-unused-function -Wno-unused-value -Wno-unused-result -Iinclude

all: $(PKG)/librt_mi355x.so oracle

# one object per source, so that a change to one file recompiles that file only (kernels.hip alone is ~25 s)
OBJDIR   := build/obj
DEVHDRS  := $(CSRC)/rt_device.h $(CSRC)/rt_kernels.h $(CSRC)/bvh_build.h $(CSRC)/bvh_gpu.h include/rt_api.h
$(OBJDIR)/%.o: $(CSRC)/%.hip $(DEVHDRS)
	@mkdir -p $(OBJDIR)
	$(HIPCC) $(HIPFLAGS) -c -o $@ $<
$(OBJDIR)/%.o: $(CSRC)/%.cpp $(DEVHDRS)
	@mkdir -p $(OBJDIR)
	$(HIPCC) $(HIPFLAGS) -c -o $@ $<
$(OBJDIR)/kernels.o: $(CSRC)/kernels_tile.inc $(CSRC)/kernels_beam.inc
PRODUCT_OBJS := $(OBJDIR)/kernels.o $(OBJDIR)/bvh_gpu.o $(OBJDIR)/rt_api.o $(OBJDIR)/bvh_build.o
$(PKG)/librt_mi355x.so: $(PRODUCT_OBJS)
	$(HIPCC) $(HIPFLAGS) -shared -o $@ $(PRODUCT_OBJS)

# the traversal alternatives that measured slower (k_packet, the quad/BVH4 kernel, 4-ary records: csrc/kernels_alt.inc) are NOT in the
# product library; `make alt` builds librt_mi355x_alt.so with them for the identity tests (RtContext(variant="alt") / RT_LIB_VARIANT=alt)
$(OBJDIR)/kernels_alt.o: $(CSRC)/kernels.hip $(CSRC)/kernels_alt.inc $(CSRC)/kernels_tile.inc $(CSRC)/kernels_beam.inc $(DEVHDRS)
	@mkdir -p $(OBJDIR)
	$(HIPCC) $(HIPFLAGS) -DRT_ALT_KERNELS -c -o $@ $(CSRC)/kernels.hip
$(PKG)/librt_mi355x_alt.so: $(OBJDIR)/kernels_alt.o $(OBJDIR)/bvh_gpu.o $(OBJDIR)/rt_api.o $(OBJDIR)/bvh_build.o
	$(HIPCC) $(HIPFLAGS) -shared -o $@ $^
alt: $(PKG)/librt_mi355x_alt.so
all: alt

oracle:
	$(MAKE) -C oracle all

# registers, scratch, LDS and occupancy of every kernel of the product TU (tests/test_host.py holds the shipped traversal kernels to their budget)
resource-usage:
	$(HIPCC) $(HIPFLAGS) -c -Rpass-analysis=kernel-resource-usage -o /dev/null $(CSRC)/kernels.hip

clean:
	rm -f $(PKG)/*.so; rm -rf $(OBJDIR); $(MAKE) -C oracle clean
.PHONY: all oracle clean resource-usage alt

# host-side library (OBJ/MTL ingest, camera, animation, stand-in mesh, JPEG decode) — g++ only
HOSTSRC := $(CSRC)/host_shim.cpp host/fly_camera.cpp host/standin.cpp host/standin_limbs.cpp $(wildcard host/jpeg_decode.cpp)
$(PKG)/librt_host.so: $(HOSTSRC) include/jpeg_decode.h include/rt_host.hpp include/obj_loader.h include/camera.h include/rt_vec.h include/config.h include/rt_api.h
	g++ -O2 -std=c++17 -fPIC -shared -pthread -Wall -Iinclude -o $@ $(HOSTSRC)
all: $(PKG)/librt_host.so

# several GPUs of one node from one host process: C++ over the C ABI + RCCL called directly (include/rt_multi.h)
$(PKG)/librt_multi.so: host/rt_multi.cpp include/rt_multi.h include/rt_api.h $(PKG)/librt_mi355x.so
	$(HIPCC) -O2 -std=c++17 -fPIC -shared -Wall -Iinclude -I/opt/rocm/include -o $@ host/rt_multi.cpp -L$(PKG) -lrt_mi355x -L/opt/rocm/lib -lrccl -Wl,-rpath,'$$ORIGIN' -Wl,-rpath,/opt/rocm/lib
all: $(PKG)/librt_multi.so

# headless C++ host (counterpart of the reference's main()); links the product libraries only
rt_headless: host/rt_headless.cpp host/fly_camera.cpp host/standin.cpp host/standin_limbs.cpp host/jpeg_decode.cpp $(PKG)/librt_mi355x.so $(PKG)/librt_multi.so include/rt_host.hpp include/rt_multi.h include/rt_api.h
	g++ -O2 -std=c++17 -pthread -Wall -Iinclude -o $@ host/rt_headless.cpp host/fly_camera.cpp host/standin.cpp host/standin_limbs.cpp host/jpeg_decode.cpp -L$(PKG) -lrt_multi -lrt_mi355x -Wl,-rpath,'$$ORIGIN/$(PKG)' -Wl,-rpath,/opt/rocm/lib -Wl,-rpath-link,/opt/rocm/lib

# kernel experiments: make exp EXP_NAME=<suffix> EXP_FLAGS="-DRT_EXP_..."  -> librt_mi355x_<suffix>.so (load with RT_LIB_VARIANT)
exp:
	$(HIPCC) $(HIPFLAGS) $(EXP_FLAGS) -shared -o $(PKG)/librt_mi355x_$(EXP_NAME).so $(CSRC)/kernels.hip $(CSRC)/bvh_gpu.hip $(CSRC)/rt_api.cpp $(CSRC)/bvh_build.cpp
