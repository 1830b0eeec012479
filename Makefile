# Builds the product library (librt_mi355x.so, gfx950), the CPU oracle (test infrastructure)
# and the command-line host.  `make -j8`.
HIPCC    ?= /opt/rocm/bin/hipcc
CXX      ?= g++
ARCH     ?= gfx950
EXTEND_WAVES ?= 4
HOSTDIR  := raytracer_amd/csrc/host
LIBDIR   := raytracer_amd/lib
OBJDIR   := build/obj

CXXFLAGS  := -O2 -fPIC -std=c++17 -Wall -Wextra -ffp-contract=off
HIPFLAGS  := -O3 -fPIC -std=c++17 --offload-arch=$(ARCH) -ffp-contract=off -Wall -Wno-unused-parameter -DRT_EXTEND_WAVES=$(EXTEND_WAVES)

HOST_SRCS := $(HOSTDIR)/json5.cpp $(HOSTDIR)/scene_config.cpp $(HOSTDIR)/obj_loader.cpp \
             $(HOSTDIR)/image_io.cpp $(HOSTDIR)/flatten.cpp $(HOSTDIR)/part_file.cpp \
             $(HOSTDIR)/progress.cpp $(HOSTDIR)/image_out.cpp $(HOSTDIR)/host_capi.cpp
HOST_OBJS := $(patsubst $(HOSTDIR)/%.cpp,$(OBJDIR)/%.o,$(HOST_SRCS))
HOST_HDRS := $(wildcard $(HOSTDIR)/*.hpp) include/rt_mi355x.h

all: $(LIBDIR)/librt_mi355x.so bin/raytracer oracle

$(OBJDIR)/%.o: $(HOSTDIR)/%.cpp $(HOST_HDRS)
	@mkdir -p $(OBJDIR)
	$(CXX) $(CXXFLAGS) -c $< -o $@

# kernels.hip is compiled once per group of instantiations (RT_PART 0-5), in parallel
KERNEL_OBJS := $(foreach p,0 1 2 3 4 5,$(OBJDIR)/kernels_$(p).o)
$(OBJDIR)/kernels_%.o: raytracer_amd/csrc/device/kernels.hip raytracer_amd/csrc/device/rt_device.h include/rt_mi355x.h
	@mkdir -p $(OBJDIR)
	$(HIPCC) $(HIPFLAGS) -DRT_PART=$* -c $< -o $@

$(OBJDIR)/bvh_build.o: raytracer_amd/csrc/device/bvh_build.hip raytracer_amd/csrc/device/rt_device.h
	@mkdir -p $(OBJDIR)
	$(HIPCC) $(HIPFLAGS) -c $< -o $@

$(OBJDIR)/capi.o: raytracer_amd/csrc/capi.cpp raytracer_amd/csrc/tree_layout.hpp raytracer_amd/csrc/device/rt_device.h $(HOST_HDRS)
	@mkdir -p $(OBJDIR)
	$(HIPCC) $(HIPFLAGS) -c $< -o $@

$(OBJDIR)/multi.o: raytracer_amd/csrc/multi.cpp raytracer_amd/csrc/device/rt_device.h include/rt_mi355x.h
	@mkdir -p $(OBJDIR)
	$(HIPCC) $(HIPFLAGS) -c $< -o $@

$(LIBDIR)/librt_mi355x.so: $(HOST_OBJS) $(KERNEL_OBJS) $(OBJDIR)/bvh_build.o $(OBJDIR)/capi.o $(OBJDIR)/multi.o
	@mkdir -p $(LIBDIR)
	$(HIPCC) -shared -fPIC --offload-arch=$(ARCH) -o $@ $^ -lz -ldl -lpthread

# Command-line host: the reference's `raytracer` binary over the C ABI.
bin/raytracer: $(HOSTDIR)/cli_main.cpp $(OBJDIR)/part_file.o $(OBJDIR)/progress.o $(OBJDIR)/image_out.o $(LIBDIR)/librt_mi355x.so
	@mkdir -p bin
	$(CXX) $(CXXFLAGS) -Iinclude -o $@ $(HOSTDIR)/cli_main.cpp $(OBJDIR)/part_file.o $(OBJDIR)/progress.o $(OBJDIR)/image_out.o \
	    -L$(LIBDIR) -lrt_mi355x -lz -lpthread -Wl,-rpath,'$$ORIGIN/../$(LIBDIR)'

oracle:
	$(MAKE) -C oracle

clean:
	rm -rf build bin $(LIBDIR)/*.so
	$(MAKE) -C oracle clean

.PHONY: all oracle clean
