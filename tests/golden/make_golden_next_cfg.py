"""Constructor arguments of the g14 golden set (tests/golden/make_golden_next.py)."""
G14 = {
    "cnn_bn_pool": ("CNN", dict(input_size=128, output_size=2, channels=4, layer_sizes=[8, 12], batch_norm=True, pool=True)),
    "cnn_groups": ("CNN", dict(input_size=64, output_size=3, channels=4, layer_sizes=[8, 16], groups=2, kernel_size=5,
                               padding=2, dilation=2)),
    "cnn_all": ("CNN", dict(input_size=96, output_size=2, channels=6, layer_sizes=[6, 12, 18], groups=3, batch_norm=True,
                            pool=True)),
    "cccnn_group": ("CCCNN", dict(input_size=64, output_size=2, channels=3, layer_sizes=[4, 6], kernel_sizes=[3, 5],
                                  padding=1, group=True)),
    "cccnn_pool": ("CCCNN", dict(input_size=80, output_size=3, channels=4, layer_sizes=[5, 5], kernel_sizes=3, pool=True)),
    "cccnn_group_pool": ("CCCNN", dict(input_size=64, output_size=2, channels=2, layer_sizes=[3], kernel_sizes=9,
                                       padding=4, group=True, pool=True)),
}
