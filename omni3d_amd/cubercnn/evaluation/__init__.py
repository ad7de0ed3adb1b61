from .omni3d_evaluation import (AnnotationIndex, Omni3DEvaluationHelper, Omni3DEvaluator, Omni3DParams, Omni3Deval,  # noqa: F401
                                box3d_overlap, box3d_overlap_groups, evaluate_groups, inference_on_dataset, instances_to_coco_json)
