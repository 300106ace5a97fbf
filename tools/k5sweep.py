import os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tools"))
from kbench import k5
for D in (17, 18, 19, 20, 21, 22, 23, 24, 39):
    k5(65536, D)
