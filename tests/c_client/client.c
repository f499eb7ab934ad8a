/* A plain C99 client of include/yolo355.h: what a maintainer's cgo / JNI / ctypes stub sees.  Built and run by
 * tests/test_c_header.py without a GPU: it only exercises entry points that must answer before any device work. */
#include <stdio.h>
#include <string.h>
#include "yolo355.h"

int main(void) {
    y355_config cfg;
    y355_engine *h = NULL;
    y355_pipeline *pl = NULL;
    int rc;
    if (y355_version() < 2) { printf("version %d\n", y355_version()); return 1; }
    memset(&cfg, 0, sizeof cfg);
    cfg.height = 100; cfg.width = 416; cfg.num_classes = 2; cfg.num_anchors = 5; cfg.max_batch = 1;
    rc = y355_create(&cfg, &h);
    if (rc != Y355_EINVAL || h != NULL) { printf("create: %d\n", rc); return 2; }
    if (strstr(y355_last_error(), "multiple of 16") == NULL) { printf("message: %s\n", y355_last_error()); return 3; }
    rc = y355_pipeline_create(&cfg, 0, 0, &pl);
    if (rc != Y355_EINVAL || pl != NULL) { printf("pipeline_create: %d\n", rc); return 4; }
    if (y355_forward(NULL, NULL, 1, 0, NULL, NULL, NULL, NULL) != Y355_EINVAL) return 5;
    if (y355_pipeline_submit(NULL, NULL, 1, 0, NULL, NULL, NULL, NULL, NULL, NULL) != Y355_EINVAL) return 6;
    printf("ok version %d handles %d\n", y355_version(), Y355_PIPE_DEFAULT_HANDLES);
    return 0;
}
