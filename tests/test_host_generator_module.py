"""fetal_net.generator: the reference's small generator helpers (generator.py:149-155, :404-419; the reference's own test_multi_class_labels)
and the names its training scripts import"""
import numpy as np

import fetal_net.generator as G


def test_names_of_the_reference_module():
    for name in ("get_training_and_validation_generators", "get_number_of_steps", "get_validation_split", "split_list", "random_list_generator",
                 "list_generator", "data_generator", "get_multi_class_labels"):
        assert callable(getattr(G, name)), name


def test_get_number_of_steps():
    assert [G.get_number_of_steps(n, 4) for n in (1, 4, 5, 8, 9)] == [1, 4, 2, 2, 3]


def test_multi_class_labels():                        # reference test/test_generator.py:58-66
    n_labels = 5
    labels = np.arange(1, n_labels + 1)
    label_map = np.asarray([[[np.arange(n_labels + 1)] * 3]])
    binary = G.get_multi_class_labels(label_map, n_labels, labels)
    assert binary.dtype == np.int8 and binary.shape == (1, n_labels, 3, n_labels + 1)
    for label in labels:
        assert np.all(binary[:, label - 1][label_map[:, 0] == label] == 1)
        assert binary[:, label - 1].sum() == (label_map[:, 0] == label).sum()
    assert np.array_equal(G.get_multi_class_labels(label_map, n_labels), binary)
    odd = G.get_multi_class_labels(label_map * 10, 2, labels=(30, 10))
    assert np.array_equal(odd[0, 0], label_map[0, 0] == 3) and np.array_equal(odd[0, 1], label_map[0, 0] == 1)
