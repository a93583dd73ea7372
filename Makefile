# Convenience targets; the driver-facing entry points are __graft_entry__.build()/smoke() and bench.py.
all:
	python -c "import __graft_entry__ as g; g.build()"

test-cpu: all
	python -m pytest tests -x -q -m "not gpu"

test-gpu: all
	python -m pytest tests -x -q -m gpu

bench: all
	python bench.py

clean:
	$(MAKE) -C neutral_amd clean
	$(MAKE) -C oracle clean
	rm -rf integration/_dropin

.PHONY: all test-cpu test-gpu bench clean
