# Convenience targets (the driver uses __graft_entry__.build(), pytest and bench.py directly).
.PHONY: build test-cpu test-gpu bench golden clean
build:            ## libsdr_amd/libsdrhip.so (hipcc, gfx950) + the test-side oracle (+ oracle/_ref where /root/reference exists)
	python -c "import __graft_entry__ as g; g.build()"
test-cpu: build   ## oracle vs golden vectors, C++ core, ABI, 2-rank gloo — no GPU needed
	python -m pytest tests -q -m "not gpu"
test-gpu: build   ## parity through the C ABI on an MI355X
	python -m pytest tests -q -m gpu
bench: build      ## one JSON line: Msamples/s, roofline, cpu_baseline
	python bench.py
golden:           ## re-cut tests/golden/ from the compiled reference (build container only)
	$(MAKE) -C oracle ref golden
clean:
	$(MAKE) -C libsdr_amd/csrc clean
	$(MAKE) -C oracle clean
	rm -rf tests/_build
