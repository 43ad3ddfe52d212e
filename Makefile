# Builds the gfx950 C-ABI library in-tree (the .so travels to the GPU box, it is git-ignored).
HIPCC ?= /opt/rocm/bin/hipcc
ARCH  ?= gfx950
CSRC  := speakerguard_amd/csrc
OBJ   := build/obj
SRCS  := $(wildcard $(CSRC)/*.hip)
OBJS  := $(patsubst $(CSRC)/%.hip,$(OBJ)/%.o,$(SRCS))
LIB   := speakerguard_amd/libspeakerguard_hip.so
FLAGS := --offload-arch=$(ARCH) -O3 -std=c++17 -fPIC -Wall -Wno-unused-function

ORACLE_SO := oracle/libconv_chain.so

# the checker's C part is built along (test infrastructure), but a host without gcc / OpenMP / -mfma must not fail the
# PRODUCT build because of it: oracle/conv_chain.py builds it lazily on first use anyway
all: $(LIB)
	-@$(MAKE) --no-print-directory oracle

# the CPU checker's C part (test infrastructure: never linked into the product library)
$(ORACLE_SO): oracle/conv_chain.c
	gcc -O2 -mfma -fopenmp -ffp-contract=off -shared -fPIC $< -o $@ -lm

oracle: $(ORACLE_SO)

$(OBJ)/%.o: $(CSRC)/%.hip $(CSRC)/sg_internal.h $(CSRC)/fft512.h $(CSRC)/loss_device.h include/speakerguard_hip.h
	@mkdir -p $(OBJ)
	$(HIPCC) $(FLAGS) -c $< -o $@

$(LIB): $(OBJS)
	$(HIPCC) --offload-arch=$(ARCH) -shared -fPIC -o $@ $(OBJS)

clean:
	rm -rf build $(LIB) $(ORACLE_SO)

.PHONY: all clean oracle
