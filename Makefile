# Builds the product library (HIP kernels + C-ABI + C++ prover host) for gfx950 and the test oracle.
#   make            -> icicle-snark_amd/lib/libicicle_snark_hip.so  (+ the three reference DSO names as symlinks)
#   make oracle     -> oracle/libbn254_oracle.so and, when /root/reference exists, oracle/_ref/libicicle_ref.so
HIPCC   ?= /opt/rocm/bin/hipcc
ARCH    ?= gfx950
PKG     := icicle-snark_amd
SRC     := $(PKG)/csrc
LIBDIR  := $(PKG)/lib
OBJDIR  := build/obj
EXTRA ?=
CXXFLAGS := -O3 -std=c++17 -fPIC -fvisibility=hidden --offload-arch=$(ARCH) -Iinclude -I$(SRC) -Wno-unused-result $(EXTRA)

SRCS := $(SRC)/runtime.cpp $(SRC)/host_ffi.cpp $(SRC)/vec_ops.hip $(SRC)/ntt.hip $(SRC)/msm_sort.hip $(SRC)/msm_g1.hip $(SRC)/msm_g2.hip $(SRC)/msm_g2_acc.hip $(SRC)/microbench.hip $(wildcard $(SRC)/prover/*.cpp) $(wildcard $(SRC)/prover/*.hip)
OBJS := $(patsubst $(SRC)/%,$(OBJDIR)/%.o,$(SRCS))
HDRS := $(wildcard $(SRC)/*.h) $(wildcard $(SRC)/prover/*.h) include/icicle_snark_hip.h $(wildcard include/*.h)

LIB := $(LIBDIR)/libicicle_snark_hip.so

RCCL_LIB := $(LIBDIR)/libicicle_snark_rccl.so

all: $(LIB) links prove dropin $(RCCL_LIB)

$(OBJDIR)/%.o: $(SRC)/% $(HDRS)
	@mkdir -p $(dir $@)
	$(HIPCC) $(CXXFLAGS) -x hip -c $< -o $@

$(LIB): $(OBJS)
	@mkdir -p $(LIBDIR)
	$(HIPCC) --offload-arch=$(ARCH) -shared -fPIC -o $@ $(OBJS) -lpthread

# the reference's Rust build links libicicle_device / libicicle_field_bn254 / libicicle_curve_bn254
# (wrappers/rust/icicle-runtime/build.rs:52, icicle-bn254/build.rs:59-60): one library, three names.
links: $(LIB)
	@cd $(LIBDIR) && for n in icicle_device icicle_field_bn254 icicle_curve_bn254; do ln -sf libicicle_snark_hip.so lib$$n.so; done

# multi-GPU exchange step (RCCL all-gather of the partial commitments); separate DSO
$(RCCL_LIB): $(SRC)/comm/rccl_comm.cpp
	@mkdir -p $(LIBDIR)
	$(HIPCC) -O2 -std=c++17 -fPIC -fvisibility=hidden --offload-arch=$(ARCH) -shared -o $@ $< -L/opt/rocm/lib -lrccl

prove: $(LIB)
	@if [ -f $(SRC)/prover/cli_main.cc ]; then \
	  $(HIPCC) -O2 -std=c++17 -Iinclude -o $(LIBDIR)/prove $(SRC)/prover/cli_main.cc -L$(LIBDIR) -licicle_snark_hip -Wl,-rpath,'$$ORIGIN'; fi

# the reference's Rust host restated call for call over the C ABI (only icicle_* / bn254_* exports): drop-in sequence
# test and timing (tests/test_dropin_sequence.py, bench.py)
dropin: $(LIB)
	$(HIPCC) -O2 -std=c++17 -Iinclude -o $(LIBDIR)/dropin_host $(SRC)/tools/dropin_host.cc -L$(LIBDIR) -licicle_snark_hip -lpthread -Wl,-rpath,'$$ORIGIN'

oracle:
	$(MAKE) -C oracle
	$(MAKE) -C oracle ref

clean:
	rm -rf build $(LIBDIR)/*.so $(LIBDIR)/prove $(LIBDIR)/dropin_host

.PHONY: all links prove dropin oracle clean
