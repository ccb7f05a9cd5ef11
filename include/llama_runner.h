/* llama_runner.h -- C mirror of the reference's bridge object and event stream, so the behaviour
 * of `LlamaRunner.run(with:config:...)` can be exercised (and tested) without a Swift / Objective-C
 * toolchain.  Names follow the reference:
 *
 *   _LlamaRunnerBridge        Sources/llamaObjCxx/headers/LlamaRunnerBridge.h:16-27
 *   _LlamaRunnerBridgeConfig  Sources/llamaObjCxx/headers/LlamaRunnerBridgeConfig.h:13-18
 *   _LlamaEvent (6 cases)     Sources/llamaObjCxx/headers/LlamaEvent.h:13-25
 *   LlamaErrorDomain / codes  Sources/llamaObjCxx/headers/LlamaError.h:12-19, LlamaError.m:10
 *   -[LlamaPredictOperation main]  Sources/llamaObjCxx/bridge/LlamaPredictOperation.mm:768-901
 *
 * Differences, all forced by the missing Apple runtime: events are delivered by a synchronous C
 * callback on the calling thread instead of dispatch_async onto a queue (.mm:903-910), and `run`
 * returns when the generation has completed instead of enqueueing an NSOperation
 * (LlamaRunnerBridge.mm:28-47).
 */
#ifndef LLAMA_RUNNER_H
#define LLAMA_RUNNER_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif
/* The library is built with -fvisibility=hidden: the declarations of this header are its whole dynamic symbol table. */
#if defined(__GNUC__)
#pragma GCC visibility push(default)
#endif

#define LLAMA_ERROR_DOMAIN "com.alexrozanski.llama.error"   /* LlamaError.m:10 */

typedef enum llama_event_type {
    LLAMA_EVENT_STARTED_LOADING_MODEL = 0,
    LLAMA_EVENT_FINISHED_LOADING_MODEL = 1,
    LLAMA_EVENT_STARTED_GENERATING_OUTPUT = 2,
    LLAMA_EVENT_OUTPUT_TOKEN = 3,
    LLAMA_EVENT_COMPLETED = 4,
    LLAMA_EVENT_FAILED = 5,
} llama_event_type;

/* _LlamaRunnerBridgeConfig + the harness-only extensions SURVEY.md section 5 routes around the
 * Swift surface (n_ctx is hard-coded to 512 in the reference, .mm:790; greedy = argmax, fact 8). */
typedef struct llama_runner_config {
    uint32_t numberOfThreads;     /* LlamaRunner.Config.numThreads, default 8  (LlamaRunner.swift:17) */
    uint32_t numberOfTokens;      /* LlamaRunner.Config.numTokens,  default 512 */
    const char *reversePrompt;    /* tokenized and then unused by the reference (.mm:815) */
    int32_t n_ctx;                /* extension: 0 = 512 */
    int32_t greedy;               /* extension: 1 = argmax instead of top-k/top-p sampling */
    int32_t seed;                 /* gpt_params.seed, default -1 (utils.h:16) */
    int32_t keepModel;            /* extension: 1 = the bridge keeps the loaded model between
                                     runs instead of the reference's load + free per run (.mm:790, :900); the
                                     loading events are still posted.  The model is freed with the bridge. */
} llama_runner_config;

/* text = token bytes for OUTPUT_TOKEN, the NSLocalizedDescription message for FAILED, else NULL;
 * code = LlamaErrorCode for FAILED (-1000 load, -1001 predict), else 0 */
typedef void (*llama_event_handler)(void *user, llama_event_type type, const char *text, uint32_t text_len, int32_t code);

typedef struct llama_runner_bridge llama_runner_bridge;

llama_runner_bridge *llama_runner_bridge_new(const char *model_path);          /* -initWithModelPath: */
void llama_runner_bridge_free(llama_runner_bridge *b);
const char *llama_runner_bridge_model_path(const llama_runner_bridge *b);      /* @property modelPath */
int64_t llama_runner_bridge_loads(const llama_runner_bridge *b);              /* model loads performed so far (keepModel tests) */

/* -runWithPrompt:config:eventHandler:eventHandlerQueue:  -- one load + one generation per call,
 * exactly like one LlamaPredictOperation (unless config->keepModel).  Runs on one bridge are
 * serialised (the reference's operation queue runs one operation at a time, LlamaRunnerBridge.mm:18-26).
 * Returns 0 if `completed` was emitted, else the error code. */
int32_t llama_runner_bridge_run(llama_runner_bridge *b, const char *prompt, const llama_runner_config *config,
                                llama_event_handler handler, void *user);

void llama_runner_config_default(llama_runner_config *c);                       /* Config.default */

#if defined(__GNUC__)
#pragma GCC visibility pop
#endif
#ifdef __cplusplus
}
#endif
#endif
