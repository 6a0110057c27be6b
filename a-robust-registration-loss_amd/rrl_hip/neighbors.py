"""Pseudo-triangle builder (reference: Sample_neighs, code/loss.py:473-485 with
utils.farthest_point_sample, code/utils.py:275-296).  SURVEY.md §8f row 1 ("next")."""


def sample_neighs(points, num_sample=5000, num_neigh=3):
    raise NotImplementedError(
        "Sample_neighs is preprocessing outside the timed path and is scheduled after the hot "
        "path (SURVEY.md §8f row 1); build pseudo-triangles with rrl_hip.synth.knn_triangles")
