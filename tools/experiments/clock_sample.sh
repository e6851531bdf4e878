# engine clock and power while the bench runs (rocm-smi sampled every 2 s next to a 6-pass bench)
mkdir -p gpurun_out/clk
python bench.py --no-cpu-baseline --no-natural-leg --steps 8 > gpurun_out/clk/bench.json 2> gpurun_out/clk/bench.err &
BP=$!
for i in $(seq 1 60); do
  if ! kill -0 $BP 2>/dev/null; then break; fi
  echo "t=$i $(rocm-smi --showclocks --showpower 2>/dev/null | grep -E 'sclk|Power' | tr -s ' ' | tr '\n' '|')" >> gpurun_out/clk/samples.txt
  sleep 2
done
wait $BP
tail -12 gpurun_out/clk/samples.txt
