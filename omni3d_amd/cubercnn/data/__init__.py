from .build import build_detection_test_loader, build_detection_train_loader, get_detection_dataset_dicts  # noqa: F401
from .dataset_mapper import DatasetMapper3D, annotations_to_instances, transform_instance_annotations  # noqa: F401
from .datasets import (Omni3D, get_filter_settings_from_cfg, get_omni3d_categories, load_omni3d_json,  # noqa: F401
                       register_and_store_model_metadata, simple_register)
