# Convenience targets; the driver-facing entry points are __graft_entry__.build()/smoke() and bench.py.
all:
	python -c "import __graft_entry__ as g; g.build()"

# the whole CPU suite, the five-minute scatter known answer included
test-cpu: all
	NEUTRAL_FULL_KATS=1 python -m pytest tests -x -q -m "not gpu" 2>&1 | tee oracle/pins/full_kats.log

test-gpu: all
	python -m pytest tests -x -q -m gpu

bench: all
	python bench.py

clean:
	$(MAKE) -C neutral_amd clean
	$(MAKE) -C oracle clean
	rm -rf integration/_dropin

.PHONY: all test-cpu test-gpu bench clean
