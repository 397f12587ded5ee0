'use strict';
// src/fps-counter.ts:1-21 with perf_hooks instead of the browser's performance object.
const { performance } = require('perf_hooks');

class FPSCounter {
    constructor() {
        this.frameCount = 0;
        this.lastTime = performance.now();
        this.currentFPS = 0;
    }

    updateFPS() {
        this.frameCount++;
        const currentTime = performance.now();
        const deltaTime = currentTime - this.lastTime;
        if (deltaTime >= 1000) {
            this.currentFPS = Math.round((this.frameCount * 1000) / deltaTime);
            this.frameCount = 0;
            this.lastTime = currentTime;
        }
    }

    getFPS() {
        return this.currentFPS;
    }
}

module.exports = { FPSCounter };
