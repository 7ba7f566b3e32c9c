V=miniworld_amd/csrc/_variants
echo "== product"; bash tools/perf/ab.sh maze 2
echo "== occ6"; MW_ENGINE_LIB=$V/libmwengine_k6.so bash tools/perf/ab.sh maze 2
python -m pytest tests -m gpu -x -q -k "render_parity or reference_gl or full_size" 2>&1 | tail -2
