# Plain-make entry points for a maintainer who integrates libmi355zk.so without the Python tooling.
#   make lib        libmi355zk.so for gfx950: four translation units (csrc/lib_*.hip), `make -j4 lib` builds them in parallel (~1 min)
#   make oracle     the CPU oracle (test infrastructure only)
#   make test-cpu   the GPU-less test suite          make test-gpu   the -m gpu suite (needs an MI355X)
#   make bench      the headline measurement (one JSON line)
HIPCC ?= /opt/rocm/bin/hipcc
CSRC := scroll-prover_amd/csrc
LIB := scroll-prover_amd/libmi355zk.so
DEPS := $(wildcard $(CSRC)/*.hpp $(CSRC)/*.hpp $(CSRC)/*.inc) include/mi355zk.h

.PHONY: lib oracle test-cpu test-gpu bench clean
OBJDIR := scroll-prover_amd/build
UNITS := lib_core lib_msm lib_ntt lib_aux
OBJS := $(UNITS:%=$(OBJDIR)/%.o)
lib: $(LIB)
$(OBJDIR)/%.o: $(CSRC)/%.hip $(DEPS)
	@mkdir -p $(OBJDIR)
	$(HIPCC) --offload-arch=gfx950 -O3 -std=c++17 -fPIC -DNDEBUG -Wno-unused-result -c $< -o $@
$(LIB): $(OBJS)
	$(HIPCC) --offload-arch=gfx950 -shared -fPIC -o $@ $(OBJS) -ldl -lpthread
oracle:
	$(MAKE) -C oracle
test-cpu: lib oracle
	python -m pytest tests -q -m "not gpu"
test-gpu: lib oracle
	python -m pytest tests -q -m gpu
bench: lib oracle
	python bench.py
clean:
	rm -rf $(LIB) $(LIB).srchash $(OBJDIR); $(MAKE) -C oracle clean
