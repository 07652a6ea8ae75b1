"""`marius_eval <config.yaml>` — evaluate the model directory written by marius_train (reference: src/cpp/src/marius.cpp:165-184)."""
import sys

from .marius_train import main

if __name__ == "__main__":
    sys.exit(main(train=False))
