# Builds the reference's own CPU library from the sources where they lie under
# /root/reference (read-only), into oracle/_ref/ only.  TEST INFRASTRUCTURE: this
# is the checker the oracle restatement is pinned against, never the product.
#
# Recipe = SURVEY.md §8(c): the 19 translation units of the reference Makefile's
# LIBOBJ list (Makefile:26-32), its own flags (makefile.inc), BLAS = MKL LP64 by
# full path (the reference names no BLAS vendor; MKL is what this image has).
# Nothing is copied: objects, the shared library and the fixture driver are the
# only outputs, and oracle/_ref/ is git-ignored.
#
#   make -f oracle/ref.mk            # from the repo root
#
REF      ?= /root/reference
OUT      := oracle/_ref
MKLDIR   ?= /opt/conda/lib

CXX      := g++
CXXFLAGS := -std=c++11 -fPIC -m64 -O3 -mavx -msse4 -mpopcnt -fopenmp \
            -Wno-sign-compare -w -DFINTEGER=int

TUS := hamming utils IndexFlat IndexIVF IndexLSH IndexPQ IndexIVFPQ Clustering \
       Heap VectorTransform index_io PolysemousTraining MetaIndexes Index \
       ProductQuantizer AutoTune AuxIndexStructures IndexScalarQuantizer \
       FaissException
OBJS := $(addprefix $(OUT)/obj/,$(addsuffix .o,$(TUS)))

MKL_LINK := -Wl,--no-as-needed $(MKLDIR)/libmkl_gf_lp64.so \
            $(MKLDIR)/libmkl_gnu_thread.so $(MKLDIR)/libmkl_core.so \
            -lgomp -lpthread -lm -ldl

all: $(OUT)/libfaiss_ref.so $(OUT)/ref_driver $(OUT)/mkl/.ok

$(OUT)/obj/%.o: $(REF)/%.cpp
	@mkdir -p $(OUT)/obj
	$(CXX) $(CXXFLAGS) -c $< -o $@

$(OUT)/libfaiss_ref.so: $(OBJS)
	$(CXX) -shared -fopenmp -o $@ $^ $(MKL_LINK)

# our own driver (oracle/ref_driver.cpp) calling the reference's public API
$(OUT)/ref_driver: oracle/ref_driver.cpp $(OUT)/libfaiss_ref.so
	$(CXX) $(CXXFLAGS) -I$(REF) -o $@ $< $(OUT)/libfaiss_ref.so \
	    -Wl,-rpath,'$$ORIGIN' $(MKL_LINK)

# run-time directory of MKL symlinks (do not put /opt/conda/lib itself on
# LD_LIBRARY_PATH: its libstdc++ is older than the system one)
$(OUT)/mkl/.ok:
	@mkdir -p $(OUT)/mkl
	for f in gf_lp64 gnu_thread sequential core def avx2 avx512 mc3; do \
	  for s in so so.1; do \
	    [ -e $(MKLDIR)/libmkl_$$f.$$s ] && ln -sf $(MKLDIR)/libmkl_$$f.$$s $(OUT)/mkl/ ; \
	  done; done; true
	touch $@

clean:
	rm -rf $(OUT)

.PHONY: all clean
