// llamatest.cpp -- command-line harness mirroring the reference's llamaTest/main.swift:23-60
// (interactive prompt loop over LlamaRunner.run, tokens printed as they stream, state changes
// reported) on top of the C mirror of the bridge (include/llama_runner.h).
//
//   llamatest MODEL_PATH [--tokens N] [--threads T] [--n_ctx C] [--greedy] [--prompt "text"]
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <iostream>
#include <string>

#include "../../../include/llama_runner.h"

static void on_event(void *, llama_event_type type, const char *text, uint32_t len, int32_t code) {
    switch (type) {
        case LLAMA_EVENT_STARTED_LOADING_MODEL:     fprintf(stderr, "[initializing]\n"); break;          // RunState.initializing
        case LLAMA_EVENT_FINISHED_LOADING_MODEL:    break;
        case LLAMA_EVENT_STARTED_GENERATING_OUTPUT: fprintf(stderr, "[generating output]\n"); break;     // RunState.generatingOutput
        case LLAMA_EVENT_OUTPUT_TOKEN:              fwrite(text, 1, len, stdout); fflush(stdout); break;
        case LLAMA_EVENT_COMPLETED:                 fprintf(stdout, "\n"); fprintf(stderr, "[completed]\n"); break;
        case LLAMA_EVENT_FAILED:                    fprintf(stderr, "[failed] %s error %d: %.*s\n", LLAMA_ERROR_DOMAIN, code, (int) len, text); break;
    }
}

int main(int argc, char **argv) {
    if (argc < 2) { fprintf(stderr, "usage: llamatest MODEL_PATH [--tokens N] [--threads T] [--n_ctx C] [--greedy] [--prompt TEXT]\n"); return 2; }
    llama_runner_config cfg;
    llama_runner_config_default(&cfg);
    std::string prompt;
    bool have_prompt = false;
    for (int i = 2; i < argc; i++) {
        const std::string k = argv[i];
        if (k == "--tokens" && i + 1 < argc) cfg.numberOfTokens = (uint32_t) atoi(argv[++i]);
        else if (k == "--threads" && i + 1 < argc) cfg.numberOfThreads = (uint32_t) atoi(argv[++i]);
        else if (k == "--n_ctx" && i + 1 < argc) cfg.n_ctx = atoi(argv[++i]);
        else if (k == "--greedy") cfg.greedy = 1;
        else if (k == "--prompt" && i + 1 < argc) { prompt = argv[++i]; have_prompt = true; }
        else { fprintf(stderr, "unknown option %s\n", k.c_str()); return 2; }
    }
    llama_runner_bridge *bridge = llama_runner_bridge_new(argv[1]);
    int rc = 0;
    if (have_prompt) {
        rc = llama_runner_bridge_run(bridge, prompt.c_str(), &cfg, on_event, nullptr);
    } else {
        for (;;) {                                     // main.swift:23-60: read a prompt, run, repeat
            fprintf(stderr, "Enter prompt: ");
            if (!std::getline(std::cin, prompt)) break;
            if (prompt.empty()) continue;
            rc = llama_runner_bridge_run(bridge, prompt.c_str(), &cfg, on_event, nullptr);
        }
    }
    llama_runner_bridge_free(bridge);
    return rc == 0 ? 0 : 1;
}
