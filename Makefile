# Builds lib/libhprlp.so (the drop-in boundary, reference name and place: <root>/lib/libhprlp.so,
# reference Makefile:114-115) for gfx950 with hipcc, plus bin/solve_mps_file.
HIPCC ?= /opt/rocm/bin/hipcc
ARCH ?= gfx950
CSRC := hpr-lp-c_amd/csrc
CXXFLAGS := -O3 -std=c++17 -fPIC -Iinclude -I$(CSRC) -ffp-contract=off -Wall -Wno-unused-result -Wno-return-type-c-linkage
HIPFLAGS := --offload-arch=$(ARCH)
BUILD := build

HIP_SRCS := $(wildcard $(CSRC)/*.hip)
CPP_SRCS := $(wildcard $(CSRC)/*.cpp)
OBJS := $(patsubst $(CSRC)/%.hip,$(BUILD)/%.hip.o,$(HIP_SRCS)) $(patsubst $(CSRC)/%.cpp,$(BUILD)/%.o,$(CPP_SRCS))
HDRS := $(wildcard $(CSRC)/*.h) $(wildcard include/*.h)

all: no-debug-variants lib/libhprlp.so lib/libhprlp.a bin/solve_mps_file $(BUILD)/solve_mps_file

# The timing experiments of rounds 2-4 (HPRLP_DBG_NOFOLD / NOTILE / HALF_TILES / NOBARRIER ...: compile-time variants of the hot
# kernels that give WRONG results by design) are no longer part of the product sources: their measurements are in profiles/ and
# HISTORY.md, their code in the history (tree of commit b4383b4).  The build refuses sources or flags that bring one back.
no-debug-variants:
	@if grep -rn "HPRLP_DBG_" $(CSRC) include; then echo "error: HPRLP_DBG_* timing variants do not belong in the product sources"; exit 1; fi
	@case "$(CXXFLAGS) $(DEFS)" in *HPRLP_DBG_*) echo "error: HPRLP_DBG_* in the compiler flags"; exit 1;; esac

$(BUILD)/%.hip.o: $(CSRC)/%.hip $(HDRS)
	@mkdir -p $(BUILD)
	$(HIPCC) $(CXXFLAGS) $(HIPFLAGS) -x hip -c $< -o $@

$(BUILD)/%.o: $(CSRC)/%.cpp $(HDRS)
	@mkdir -p $(BUILD)
	$(HIPCC) $(CXXFLAGS) $(HIPFLAGS) -x hip -c $< -o $@

lib/libhprlp.so: $(OBJS)
	@mkdir -p lib
	$(HIPCC) -shared -fPIC $(HIPFLAGS) -o $@ $(OBJS) -lz -ldl -Wl,--no-undefined

# the reference's other two artefacts at their reference paths (reference Makefile:114,118): the static archive
# (link it with hipcc: the objects carry gfx950 code) and build/solve_mps_file
lib/libhprlp.a: $(OBJS)
	@mkdir -p lib
	rm -f $@ && ar rcs $@ $(OBJS)

$(BUILD)/solve_mps_file: bin/solve_mps_file
	cp $< $@

bin/solve_mps_file: tools/solve_mps_file.cpp lib/libhprlp.so include/HPRLP.h
	@mkdir -p bin
	g++ -O2 -std=c++11 -Iinclude -o $@ tools/solve_mps_file.cpp -Llib -lhprlp -Wl,-rpath,'$$ORIGIN/../lib'

# developer variants of the library (kernel experiments, A/B on one GPU box through HPRLP_LIB=lib/variants/libhprlp_<NAME>.so):
#   make variant NAME=ed3 DEFS="-DHPRLP_SWEEP_ED=3"
variant:
	$(MAKE) BUILD=build_variants/$(NAME) CXXFLAGS='$(CXXFLAGS) $(DEFS)' build_variants/$(NAME)/libhprlp.so
	@mkdir -p lib/variants
	cp build_variants/$(NAME)/libhprlp.so lib/variants/libhprlp_$(NAME).so

$(BUILD)/libhprlp.so: $(OBJS)
	$(HIPCC) -shared -fPIC $(HIPFLAGS) -o $@ $(OBJS) -lz -ldl -Wl,--no-undefined

clean:
	rm -rf $(BUILD) lib bin
.PHONY: all clean variant no-debug-variants
