import sys, os; sys.path.insert(0, os.getcwd())
import bench_encoder as b, torch
d=torch.device("cuda:0")
r=b.bench_cost_volume(d, 20, 3, cpu=False); print("native", round(r["value"]), round(r["roofline"]["frac"],4), round(r["roofline"]["avg_launch_ms"],4))
r=b.bench_cost_volume(d, 5, 2, V=3, K=2, h4=242, w4=324, cpu=False); print("c3", round(r["value"]), round(r["roofline"]["frac"],4), round(r["roofline"]["avg_launch_ms"],4))
