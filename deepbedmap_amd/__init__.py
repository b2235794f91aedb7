"""deepbedmap_amd -- MI355X-native (gfx950 HIP) drop-in for the ESRGAN hot path of weiji14/deepbedmap.

Mirrors the hot-path interface of the reference's srgan_train.py; every numeric call goes through
libdbm.so (C ABI: include/dbm.h).  There is no CPU fallback: without the built library and an
MI355X the first model/context creation raises.
"""
from ._lib import DbmError, Context, default_context, build  # noqa: F401
from .srgan import (  # noqa: F401
    Adam, DeepbedmapInputBlock, DeviceArray, DiscriminatorModel, GeneratorModel, ResidualDenseBlock,
    ResInResDenseBlock, Variable, calculate_discriminator_loss, calculate_generator_loss, config, global_config,
    infer_num_residual_blocks, load_npz, load_trained_model, optimizers, psnr, save_npz, serializers, ssim_loss_func, to_device, using_config,
)
from .training import (  # noqa: F401
    METRIC_NAMES, MetricsLog, SerialIterator, TrialPruned, compile_srgan_model, concat_examples, dataset_to_device, device_batch, get_train_dev_iterators,
    save_model_weights_and_architecture, split_dataset_random, train_epochs, train_eval_discriminator, train_eval_generator, train_iteration, train_minibatch, trainer,
)
from .parallel import DataParallel, shard_batch, shard_slice  # noqa: F401
from .geotiff import canvas_to_int16, read_geotiff, save_array_to_grid  # noqa: F401
from .inference import (Shape, clip_inputs, crop_bounds, group_tiles_by_crop_shape, merge_ranks, predict_tiled,  # noqa: F401
                        predict_tiled_resident, tile_steps)
