"""cProfile of the host side of the video path (tools/video_bench.py's run): where the wall time of a tracked frame goes."""
import cProfile
import os
import pstats
import sys

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import video_bench

pr = cProfile.Profile()
pr.enable()
video_bench.run("large", 32, reps=2)
pr.disable()
pstats.Stats(pr).sort_stats("cumulative").print_stats(45)
